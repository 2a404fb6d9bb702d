"""CPU: the SVM oracle (float32 kernel sum, libsvm's label and probability routines) against LIBSVM ITSELF
(tests/golden/svm_libsvm_ref.npz: sklearn.svm._libsvm fed the attributes decoded from cepstrum/scrubjay_svm.onnx,
tools/pin_svm_libsvm.py) and against an independent float64 numpy evaluation of the decision function.
Pinned: decision value, sign convention, vote label, predict_proba, and the polarity (1 = scrub jay) on the
reference's 13 labelled WAVs.  Still unpinned: ONNX Runtime's float32 rounding of the same routines (onnxruntime is
not in this image) and the aubio front end of scrubjay_infer.c."""
import numpy as np

from oracle import oracle as O
from tests import signals as S


def _f64(model, x):
    z = (x.astype(np.float64) - model["offset"]) * model["scale"]
    d2 = ((z[None, :] - model["sv"].astype(np.float64)) ** 2).sum(1)
    score = float((model["coef"].astype(np.float64) * np.exp(-float(model["kernel_params"][0]) * d2)).sum() + float(model["rho"][0]))
    return score


def test_decoded_graph_shape(golden):
    m = golden("scrubjay_svm.npz")
    assert m["sv"].shape == (55, 40) and m["coef"].shape == (55,) and list(m["vectors_per_class"]) == [26, 29]
    assert str(m["kernel_type"]) == "RBF" and abs(float(m["kernel_params"][0]) - 0.025) < 1e-7
    assert abs(float(m["rho"][0]) - 0.1593) < 1e-3 and abs(float(m["prob_a"][0]) + 2.2436) < 1e-3


def test_oracle_matches_float64_semantics(golden):
    m = golden("scrubjay_svm.npz")
    model = {k: m[k] for k in ("offset", "scale", "sv", "coef")}
    model.update(gamma=float(m["kernel_params"][0]), rho=float(m["rho"][0]), prob_a=float(m["prob_a"][0]), prob_b=float(m["prob_b"][0]))
    labels = []
    for i in range(200):
        # feature vectors scattered around the training distribution (offset +- a few 1/scale)
        x = (m["offset"] + S.uniform_pm1(40, 900 + i) * (2.5 / m["scale"])).astype(np.float32)
        lab, dec, p1 = O.svm_predict(model, x)
        want_dec = _f64(m, x)
        assert abs(dec - want_dec) <= 2e-5 * max(1.0, abs(want_dec))
        if abs(want_dec) > 1e-5:
            assert lab == int(want_dec <= 0)          # svm_predict: a positive decision value votes for class 0
        labels.append(lab)
    assert 0 < sum(labels) < 200          # both classes occur


def _model(m):
    model = {k: m[k] for k in ("offset", "scale", "sv", "coef")}
    model.update(gamma=float(m["kernel_params"][0]), rho=float(m["rho"][0]), prob_a=float(m["prob_a"][0]), prob_b=float(m["prob_b"][0]))
    return model


def test_oracle_matches_libsvm(golden):
    m, r = golden("scrubjay_svm.npz"), golden("svm_libsvm_ref.npz")
    model = _model(m)
    loose = 0
    for i in range(r["feat"].shape[0]):
        lab, dec, p1 = O.svm_predict(model, r["feat"][i])
        assert abs(dec - r["decision"][i]) <= 2e-5
        if abs(r["decision"][i]) > 1e-5:
            assert lab == int(r["label_vote"][i])
        d = abs(p1 - r["proba"][i, 1])
        assert d <= 2.6e-3                                # the iteration's tolerance (a stopping test that falls the other way)
        loose += d > 2e-5
    assert loose <= 3
    sliver = (r["decision"] > 1e-5) & (r["decision"] < 0.0079)
    assert sliver.sum() >= 16                             # the zone where "sigmoid + arg max" would answer 1 and libsvm answers 0


def test_polarity_on_the_reference_labelled_wavs(golden):
    """13 labelled WAVs of cepstrum/data and cepstrum/testing, features from a float64 numpy restatement of train.py's
    librosa call (tools/pin_svm_libsvm.py): libsvm's predict equals the file label on all of them (the flipped sign
    convention would score 0 / 13), and the oracle agrees with libsvm."""
    m, r = golden("scrubjay_svm.npz"), golden("svm_libsvm_ref.npz")
    assert np.array_equal(r["labelled_vote"], r["labelled_y"]) and 0 < r["labelled_y"].sum() < r["labelled_y"].size
    model = _model(m)
    for i in range(r["labelled_feat"].shape[0]):
        lab, dec, p1 = O.svm_predict(model, r["labelled_feat"][i])
        assert lab == int(r["labelled_y"][i]) and abs(dec - r["labelled_decision"][i]) <= 2e-5 and abs(p1 - r["labelled_proba"][i, 1]) <= 2e-5

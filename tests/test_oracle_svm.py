"""CPU: the SVM oracle (float32, ORT operation order) against an independent float64
numpy evaluation of the attributes decoded from cepstrum/scrubjay_svm.onnx.
PARITY UNPINNED at the ONNX Runtime boundary (onnxruntime / aubio are not in this image);
what is pinned here is that the oracle implements the published SVMClassifier semantics."""
import numpy as np

from oracle import oracle as O
from tests import signals as S


def _f64(model, x):
    z = (x.astype(np.float64) - model["offset"]) * model["scale"]
    d2 = ((z[None, :] - model["sv"].astype(np.float64)) ** 2).sum(1)
    score = float((model["coef"].astype(np.float64) * np.exp(-float(model["kernel_params"][0]) * d2)).sum() + float(model["rho"][0]))
    f = score * float(model["prob_a"][0]) + float(model["prob_b"][0])
    p0 = 1.0 / (1.0 + np.exp(f))
    return score, 1.0 - p0


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
        want_dec, want_p1 = _f64(m, x)
        assert abs(dec - want_dec) <= 2e-5 * max(1.0, abs(want_dec))
        assert abs(p1 - want_p1) <= 2e-5
        if abs(want_p1 - 0.5) > 1e-4:
            assert lab == int(want_p1 > 0.5)
        labels.append(lab)
    assert 0 < sum(labels) < 200          # both classes occur

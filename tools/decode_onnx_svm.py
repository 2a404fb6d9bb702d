#!/usr/bin/env python3
"""Minimal protobuf reader for cepstrum/scrubjay_svm.onnx (no onnx / onnxruntime here):
pulls the attributes of the Scaler and SVMClassifier nodes into a dict of numpy arrays.

    python tools/decode_onnx_svm.py /root/reference/cepstrum/scrubjay_svm.onnx [out.npz]
"""
import struct
import sys

import numpy as np


def varint(b, i):
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def fields(b):
    i = 0
    while i < len(b):
        key, i = varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            n, i = varint(b, i)
            v = b[i:i + n]; i += n
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError(wt)
        yield f, wt, v


def attr(b):
    """AttributeProto: name=1, f=2, i=3, s=4, floats=7, ints=8, strings=9, type=20."""
    out = {"floats": [], "ints": [], "strings": []}
    for f, wt, v in fields(b):
        if f == 1: out["name"] = v.decode()
        elif f == 2: out["f"] = struct.unpack("<f", v)[0]
        elif f == 3: out["i"] = v
        elif f == 4: out["s"] = v.decode(errors="replace")
        elif f == 7:
            if wt == 2: out["floats"] += list(struct.unpack(f"<{len(v) // 4}f", v))
            else: out["floats"].append(struct.unpack("<f", v)[0])
        elif f == 8:
            if wt == 2:
                i = 0
                while i < len(v):
                    x, i = varint(v, i); out["ints"].append(x)
            else: out["ints"].append(v)
        elif f == 9: out["strings"].append(v.decode(errors="replace"))
    return out


def decode(path):
    model = open(path, "rb").read()
    graph = next(v for f, wt, v in fields(model) if f == 7)
    nodes = {}
    for f, wt, v in fields(graph):
        if f != 1:
            continue
        node = {"attrs": {}}
        for nf, nwt, nv in fields(v):
            if nf == 4: node["op"] = nv.decode()
            elif nf == 5:
                a = attr(nv); node["attrs"][a["name"]] = a
        nodes[node.get("op", "?")] = node
    return nodes


def main():
    nodes = decode(sys.argv[1])
    print("ops:", list(nodes))
    sc, svm = nodes["Scaler"]["attrs"], nodes["SVMClassifier"]["attrs"]
    for k, a in svm.items():
        desc = a.get("s") or a.get("f") or a.get("i") or (f"{len(a['floats'])} floats" if a["floats"] else a["ints"] or a["strings"])
        print(f"  {k}: {desc}")
    n_sv = sum(svm["vectors_per_class"]["ints"])
    sv = np.array(svm["support_vectors"]["floats"], np.float32).reshape(n_sv, -1)
    out = dict(
        offset=np.array(sc["offset"]["floats"], np.float32), scale=np.array(sc["scale"]["floats"], np.float32),
        sv=sv, coef=np.array(svm["coefficients"]["floats"], np.float32),
        rho=np.array(svm["rho"]["floats"], np.float32), kernel_params=np.array(svm["kernel_params"]["floats"], np.float32),
        prob_a=np.array(svm["prob_a"]["floats"], np.float32), prob_b=np.array(svm["prob_b"]["floats"], np.float32),
        vectors_per_class=np.array(svm["vectors_per_class"]["ints"], np.int32),
        classlabels=np.array(svm["classlabels_ints"]["ints"], np.int64),
        kernel_type=svm["kernel_type"]["s"], post_transform=svm["post_transform"]["s"],
    )
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], **out)
    return out


if __name__ == "__main__":
    main()

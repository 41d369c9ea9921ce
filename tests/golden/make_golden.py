"""Extract the golden vectors the reference's own tests hold for the conv/TDS
forward path into JSON fixtures (run in the build container only; the GPU box
has no /root/reference).

Sources (read-only):
  recipes/streaming_convnets/inference/inference/module/test/Conv1dTest.cpp:30-104
  recipes/streaming_convnets/inference/inference/module/test/TDSBlockTest.cpp:27-188
Only the numeric test vectors (data, not code) are extracted.
"""
import json
import os
import re

REF = "/root/reference/recipes/streaming_convnets/inference/inference/module/test"
OUT = os.path.dirname(os.path.abspath(__file__))


def grab(src, name):
    m = re.search(r"std::vector<float>\s+" + name + r"\s*=\s*\{(.*?)\};", src, re.S)
    assert m, name
    return [float(v) for v in re.findall(r"-?\d+\.?\d*(?:[eE]-?\d+)?", m.group(1))]


def main():
    src = open(os.path.join(REF, "Conv1dTest.cpp")).read()
    src = src[src.index("TEST(Conv1d, SingleLayer)"):src.index("TEST(Conv1d, SingleLayerSerialization)")]
    conv = dict(T=10, groups=5, channels=10, kernelSize=3, stride=1, leftPadding=1, rightPadding=1,
                tol=1e-2,
                input=grab(src, "inputValues"), target=grab(src, "targetValues"),
                weights=grab(src, "weightsValues"), bias=grab(src, "biasValues"),
                source="Conv1dTest.cpp:30-104")
    assert len(conv["input"]) == 100 and len(conv["target"]) == 100
    assert len(conv["weights"]) == 12 and len(conv["bias"]) == 2
    json.dump(conv, open(os.path.join(OUT, "conv1d_golden.json"), "w"), indent=0)

    src = open(os.path.join(REF, "TDSBlockTest.cpp")).read()
    src = src[src.index("TEST(TDSBlock, TestOne)"):src.index("TEST(TDSBlock, Serialization)")]
    tds = dict(T=10, groups=5, channels=10, kernelSize=3, stride=1, leftPadding=1, rightPadding=1,
               tol=1e-2, source="TDSBlockTest.cpp:27-188")
    for k in ("conv_weights", "conv_bias", "ln1_weights", "ln1_bias", "lin1_weights", "lin1_bias",
              "lin2_weights", "lin2_bias", "ln2_weights", "ln2_bias", "in", "expectedOutput"):
        tds[k] = grab(src, k)
    assert len(tds["in"]) == 100 and len(tds["expectedOutput"]) == 100
    assert len(tds["lin1_weights"]) == 100 and len(tds["lin2_weights"]) == 100
    json.dump(tds, open(os.path.join(OUT, "tdsblock_golden.json"), "w"), indent=0)
    print("wrote conv1d_golden.json, tdsblock_golden.json")


if __name__ == "__main__":
    main()

"""Extract the golden vectors the reference's own tests hold for the conv/TDS
forward path into JSON fixtures (run in the build container only; the GPU box
has no /root/reference).

Sources (read-only):
  recipes/streaming_convnets/inference/inference/module/test/Conv1dTest.cpp:30-104
  recipes/streaming_convnets/inference/inference/module/test/TDSBlockTest.cpp:27-188
  recipes/streaming_convnets/inference/inference/module/test/LinearTest.cpp:32-87 (two known-answer cases)
  recipes/streaming_convnets/inference/inference/module/test/ReluTest.cpp:27-50 (one known-answer case)
  recipes/streaming_convnets/inference/inference/module/test/LayerNormTest.cpp:31-70, ResidualTest.cpp:28-82 (the PROPERTIES
  these two assert on random data: their sizes, constants and tolerances)
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
    # ---- the other known answers the reference's module tests hold (Linear, Relu) and the properties two of them assert
    def case(text, first, last):
        return text[text.index(first):text.index(last)]

    def near(text):     # ASSERT_NEAR(out[i], value, tol) in order
        return [(float(v.replace("INT_MAX", "2147483647")), float(t)) for v, t in
                re.findall(r"ASSERT_NEAR\(out\[\d+\],\s*([-\w.]+),\s*([-+\w.]+)\)", text)]

    def vec(text, name):
        m = re.search(r"std::vector<float>\s+" + name + r"\s*=\s*\{(.*?)\};", text, re.S)
        assert m, name
        body = m.group(1).replace("static_cast<float>(INT_MAX)", "2147483647")
        return [float(v) for v in re.findall(r"-?\d+\.?\d*(?:[eE]-?\d+)?", body)]

    lin = open(os.path.join(REF, "LinearTest.cpp")).read()
    c1 = case(lin, "TEST(Linear, SingleNeuronSingleFrame)", "TEST(Linear, MultiNeuronMultiFrame)")
    c2 = case(lin, "TEST(Linear, MultiNeuronMultiFrame)", "// Same test as MultiNeuronMultiFrame")
    relu = open(os.path.join(REF, "ReluTest.cpp")).read()
    ln = open(os.path.join(REF, "LayerNormTest.cpp")).read()
    lnc = case(ln, "TEST(LayerNorm, Batch)", "TEST(LayerNorm, BatchChunked)")
    res = open(os.path.join(REF, "ResidualTest.cpp")).read()
    resc = case(res, "TEST(Residual, ConvResidual)", "TEST(Residual, ConvResidualSerialization)")

    def ints(text, *names):
        return {n: int(re.search(r"\b" + n + r"\s*=\s*(\d+)", text).group(1)) for n in names}
    known = dict(
        linear=[dict(nIn=3, nOut=1, weights=vec(c1, "weightsValues"), bias=vec(c1, "biasValues"), input=vec(c1, "inputValues"),
                     expected=near(c1), source="LinearTest.cpp:32-60"),
                dict(nIn=3, nOut=2, weights=vec(c2, "weightsValues"), bias=vec(c2, "biasValues"), input=vec(c2, "inputValues"),
                     expected=near(c2), source="LinearTest.cpp:62-87")],
        relu=dict(input=vec(relu, "inputValues"), expected=near(relu), source="ReluTest.cpp:27-50"),
        layernorm_property=dict(**ints(lnc, "T", "F"), alpha=float(re.search(r"alpha\s*=\s*([\d.]+)", lnc).group(1)),
                                beta=float(re.search(r"beta\s*=\s*([\d.]+)", lnc).group(1)),
                                tol=float(re.search(r"EXPECT_NEAR\(outPtr\[i \* F \+ j\], e, ([\de.+-]+)\)", lnc).group(1)),
                                statement="row i drawn from N(mean_i, std_i), mean / std uniform in [0, 1): out = alpha (in - mean_i) / std_i + beta",
                                source="LayerNormTest.cpp:31-70"),
        residual_property=dict(**ints(resc, "T", "groups", "channels", "kernelSize", "stride", "rightPadding", "leftPadding"),
                               input_value=1.0, tol=float(re.search(r"ASSERT_NEAR\(outNoRes\[i\] \+ 1.0, outRes\[i\], ([\dE.+-]+)\)", resc).group(1)),
                               statement="Residual(conv)(ones) = conv(ones) + 1 for a random grouped convolution", source="ResidualTest.cpp:28-82"))
    assert known["linear"][0]["expected"] == [(40.0, 1e-3)] and len(known["linear"][1]["expected"]) == 4
    assert len(known["relu"]["input"]) == 6 and len(known["relu"]["expected"]) == 6
    json.dump(known, open(os.path.join(OUT, "reference_known_answers.json"), "w"), indent=1)
    print("wrote conv1d_golden.json, tdsblock_golden.json, reference_known_answers.json")


if __name__ == "__main__":
    main()

"""Static instruction mix of the round loop of the block-Toeplitz TDS convolution kernels, read off the gfx950 ISA hipcc
emits (conv_tds_rs.hip compiled with -save-temps): MFMA / DS / VMEM / VALU / SALU instructions per round of a wave, and the
issue-model ceiling of profiles/r02_run15_mfma_issue_microbench.log (a DS instruction beside an MFMA costs the SIMD's matrix
pipe ~10 cycles, a VALU / VMEM instruction ~4.5) times the algorithmic share of the issued MFMA flops.
    python tools/conv_tz_isa_counts.py /tmp/conv_tds_rs-hip-amdgcn-amd-amdhsa-gfx950.s > profiles/r05_tds_conv_issue_model.json"""
import json
import re
import sys

src = open(sys.argv[1]).read()
out = {"_model": "cycles per MFMA = 64 + 10 DS/MFMA + 4.5 (VALU + VMEM)/MFMA over the round loop of one wave; ceiling = 64 / cycles * useful / issued flops",
       "_source": "ISA of conv_tds_rs.hip (hipcc -O3 --offload-arch=gfx950), tools/conv_tz_isa_counts.py", "kernels": {}}
useful = {  # algorithmic / issued MFMA flops: taps / S x columns / (32 NCT) x K / K padded
    "tds_conv_tz_k<10,10>": 21 / 23 * 30 / 32 * 230 / 232, "tds_conv_tz_k<14,14>": 21 / 22 * 28 / 32, "tds_conv_tz_k<18,18>": 21 / 23 * 54 / 64 * 414 / 416,
    "tds_conv_tzf_k<10,10>": 210 * 30 / (256 * 32), "tds_conv_tzf_k<14,14>": 294 * 28 / (320 * 32)}
for m in re.finditer(r"^(_ZN3w2l1[34]tds_conv_tz[f]?_kILi(\d+)ELi(\d+)E[^:\n]*):[^\n]*\n(.*?)\.end_amdhsa_kernel", src, re.S | re.M):
    name, ci, co, body = m.group(1), m.group(2), m.group(3), m.group(4)
    if ci != co:
        continue
    filt = "tzf" in name
    if not filt and not re.search(r"ELi21ELi1ELi1ELb[01]ELi0E", name):     # forward + ReLU of the stride-1 instances only
        continue
    # the round loop: the basic block(s) with the most MFMAs between a label and the backward branch
    lines = body.split("\n")
    best = None
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^\.LBB\d+_\d+:", l):
            start = i
        if start is not None and re.search(r"s_cbranch_\w+ \.LBB|s_branch \.LBB", l):
            seg = lines[start:i + 1]
            n = sum("v_mfma" in x for x in seg)
            if best is None or n > best[0]:
                best = (n, seg)
    n, seg = best
    ins = [x.strip().split()[0] for x in seg if x.startswith("\t") and not x.strip().startswith((";", "."))]
    cnt = {"mfma": sum(i.startswith("v_mfma") for i in ins), "ds": sum(i.startswith("ds_") for i in ins),
           "vmem": sum(i.startswith(("buffer_", "global_")) for i in ins),
           "valu": sum(i.startswith("v_") and not i.startswith("v_mfma") for i in ins), "salu": sum(i.startswith("s_") for i in ins)}
    key = ("tds_conv_tzf_k" if filt else "tds_conv_tz_k") + f"<{ci},{co}>"
    cyc = 64 + 10 * cnt["ds"] / cnt["mfma"] + 4.5 * (cnt["valu"] + cnt["vmem"]) / cnt["mfma"]
    out["kernels"][key] = {"per_round_of_a_wave": cnt, "ds_per_mfma": round(cnt["ds"] / cnt["mfma"], 3),
                           "valu_vmem_per_mfma": round((cnt["valu"] + cnt["vmem"]) / cnt["mfma"], 3), "model_cycles_per_mfma": round(cyc, 1),
                           "useful_frac_of_issued": round(useful[key], 3), "ceiling_frac_of_peak": round(64 / cyc * useful[key], 3)}
print(json.dumps(out, indent=1))

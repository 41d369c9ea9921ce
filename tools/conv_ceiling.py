"""Issue-model ceiling of the fp32 TDS convolution kernels, recomputable from committed evidence.

  inputs   profiles/r02_run15_conv_sq_lds_pmc.csv        SQ counters of the wave-specialised kernels at the bench shapes
                                                          (instructions per dispatch: MFMA, LDS, VALU incl. MFMA, SALU)
           profiles/r02_run15_mfma_issue_microbench.log  what an instruction beside an MFMA costs the matrix pipe of its SIMD:
                                                          scalar instructions and waits 0, a VALU instruction 3 - 6 cycles (4.5
                                                          taken), a DS instruction ~10 cycles; a v_mfma_f32_32x32x2_f32 occupies
                                                          the pipe for 64 cycles
  model    cycles per MFMA = 64 + 10 * (LDS / MFMA) + 4.5 * ((VALU - MFMA) / MFMA)
           ceiling = 157.3 TF/s * 64 / cycles per MFMA * (algorithmic flops / issued MFMA flops)
           (issued MFMA flops = MFMA instructions * 4096; the role-swapped kernels pad tap groups x channels to the 32 MFMA columns)
  output   profiles/r04_tds_conv_issue_model.json (read by bench.py: tds_conv.ceiling)

The model is an UPPER bound on what these instruction streams can reach (it ignores barriers, LDS bank conflicts, the prologue /
epilogue of a workgroup and the tail of the grid); it does not say that a leaner stream is impossible."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 157.3
# algorithmic GFLOP per launch at the bench shape (B = 32, T = 1500): conv of one TDS block, forward = backward-data =
# backward-filter = 2 * M * K * N
ALG = {10: 8.06, 14: 7.90, 18: 6.55}
lines = [l for l in open(os.path.join(ROOT, "profiles", "r02_run15_conv_sq_lds_pmc.csv")).read().splitlines() if l.strip()]
hdr = lines[0].split(",")
out = {"_model": "cycles per MFMA = 64 + 10 LDS/MFMA + 4.5 (VALU - MFMA)/MFMA; ceiling = 157.3 * 64 / cycles * algorithmic / issued flops",
       "_sources": ["profiles/r02_run15_conv_sq_lds_pmc.csv", "profiles/r02_run15_mfma_issue_microbench.log"], "kernels": {}}
for l in lines:
    if l.startswith("kernel,"):
        continue
    row = next(csv.reader([l]))
    d = dict(zip(hdr, row))
    name = d["kernel"]
    mfma = float(d["SQ_INSTS_MFMA"])
    if mfma == 0:
        continue
    C = int(name.split("<")[1].split(",")[0])
    lds, valu = float(d["SQ_INSTS_LDS"]), float(d["SQ_INSTS_VALU"]) - mfma
    cyc = 64 + 10 * lds / mfma + 4.5 * valu / mfma
    issued = mfma * 4096 / 1e9
    useful = min(1.0, ALG[C] / issued)
    key = ("tds_conv_rs3_k" if "rs3_k" in name else "tds_conv_rsf3_k") + f"<C={C}>"
    out["kernels"][key] = {"mfma_per_dispatch": mfma, "lds_per_mfma": round(lds / mfma, 3), "valu_per_mfma": round(valu / mfma, 3),
                           "model_cycles_per_mfma": round(cyc, 1), "issued_gflop": round(issued, 2), "algorithmic_gflop": ALG[C],
                           "useful_frac_of_issued": round(useful, 3), "ceiling_TFLOPs": round(PEAK * 64 / cyc * useful, 1),
                           "ceiling_frac_of_peak": round(64 / cyc * useful, 3)}
ks = out["kernels"]
out["step_weighted_ceiling_frac"] = round(sum(v["ceiling_frac_of_peak"] for v in ks.values()) / len(ks), 3)
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_tds_conv_issue_model.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

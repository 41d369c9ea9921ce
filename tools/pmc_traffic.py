#!/usr/bin/env python
"""Turn the per-kernel FETCH_SIZE / WRITE_SIZE summaries written by tools/pmc.sh into
profiles/<tag>_pmc_traffic.json: HBM-side bytes per launch of the dominant kernels.

Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3's FETCH_SIZE is in KiB and, on gfx950,
tallies the 128-byte requests of wide coalesced streaming reads at 64 bytes -- doubled here.
WRITE_SIZE (KiB) is uncalibrated on gfx950 and reported as is.
"""
import csv
import json
import sys


def load(path, col):
    out = {}
    try:
        for r in csv.DictReader(open(path)):
            out[r["kernel"]] = (int(r["dispatches"]), float(r[col]))
    except (OSError, KeyError):
        pass
    return out


def main():
    fetch_csv, write_csv, out_json = sys.argv[1:4]
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    groups = {"gemm_lds_dma": ("gemm128g_kernel", "gemm160_kernel"), "gemm128g": ("gemm128g_kernel",), "gemm160": ("gemm160_kernel",),
              "gemm128_kernel": ("gemm128_kernel",), "fcc_big_gemm": ("fcc_big_gemm",),
              "fcc_big_gemm_alpha": ("fcc_big_gemm_dma<true",), "fcc_big_gemm_beta": ("fcc_big_gemm_dma<false",),
              "attn_fused_bwd": ("attn_fused_bwd_q_k", "attn_fused_bwd_kv_k"), "ln_images": ("ln_rows_images_k",),
              "tds_conv_fwd2": ("tds_conv_fwd2_k",), "tds_conv_filter2": ("tds_conv_filter2_k",),
              "tds_conv_tz": ("tds_conv_tz_k",), "tds_conv_tzf": ("tds_conv_tzf_k",), "tds_conv_c1": ("tds_c1_fwd_k", "tds_c1_filter_k"),
              "tds_conv_rs": ("tds_conv_rs_k", "tds_conv_rs3_k"), "tds_conv_rsf": ("tds_conv_rsf_k", "tds_conv_rsf3_k"), "gemm_bf16": ("gemm128_bf16_kernel",),
              "gemm_bf16_images": ("gemm128h_kernel", "gemm256h_kernel"), "cvt_bf16": ("cvt_bf16_k", "cvt_bf16_multi_k"),
              "tds_conv_bf16": ("tds_conv_bf_k",), "tds_conv_bf16_filter": ("tds_conv_bf_filter",), "attn_fused_fwd": ("attn_fused_fwd_k",)}
    res = {}
    for key, subs in groups.items():
        n = fb = wb = 0.0
        for k, (disp, v) in f.items():
            if any(sub in k for sub in subs):
                n += disp
                fb += disp * v * 1024.0 * 2.0
        nw = 0.0
        for k, (disp, v) in w.items():
            if any(sub in k for sub in subs):
                nw += disp
                wb += disp * v * 1024.0
        if n:
            res[key] = {"launches": int(n), "fetch_bytes_per_launch": fb / n,
                        "write_bytes_per_launch": (wb / nw) if nw else None,
                        "hbm_bytes_per_launch": fb / n + ((wb / nw) if nw else 0.0)}
    res["_note"] = ("FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-B read requests at 64 B); WRITE_SIZE KiB x 1024 "
                    "uncorrected; separate --pmc passes, kernels serialised by the profiler")
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

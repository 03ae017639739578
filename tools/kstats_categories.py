"""Category table of a rocprofv3 `--kernel-trace --stats` CSV (the *_kernel_stats.csv files under profiles/):
    python tools/kstats_categories.py <kernel_stats.csv> <steps in the trace> [--top N]
ms and launches per step by kernel family - the budget lines DESIGN.md / VERDICT.md quote."""
import csv
import re
import sys

CATS = [
    ("2d conv 3x3 fwd+dgrad", r"k_conv3x3w|k_conv3x3v|k_conv3x3s|k_conv3x3r|k_c3"),
    ("2d conv wgrad (+reduce)", r"k_wgrad3x3n|k_wgrad_reduce|k_conv_wgrad2|k_wg_"),
    ("2d conv gemm/stem/dgrad_s2", r"k_conv_gemm|k_stem7|k_stem_prep|k_conv_f32|k_wgrad_f32"),
    ("BatchNorm2d", r"k_bn2d|k_colsum"),
    ("sparse engines", r"k_osconv|k_gather_gemm|k_csr_reduce|k_dw_direct|k_dw_tr16|k_dw_reduce|k_rows_narrow"),
    ("sparse batch norm", r"k_bn_"),
    ("sparse metadata (own)", r"k_insert|k_flag|k_assign|k_subm_nbr|k_down_nbr|k_up_nbr|k_emit_rules|k_row_fill|k_os_fill|k_row_mask|"
                              r"k_batch_lower|k_csr_|k_init_level|k_meta|k_hash|k_dedupe|k_scan|k_tile|k_bucket"),
    ("rocPRIM", r"rocprim"),
    ("2d misc (pool/heads/concat/pack/dropout)", r"k_maxpool|k_head|k_box5|k_concat|k_pack_weights|k_copy_rows|k_dropout|k_bnpool"),
    ("points / lifting / losses", r"k_gate|k_seg_mean|k_row_gather|k_linear|k_lift|k_ce_|k_kl_|k_segment|k_key"),
    ("optimiser / amp / packs", r"k_adamw|k_amp|k_grad_nonfinite|k_os_pack|k_pack_frag"),
    ("torch (at::)", r"at::|at_cuda|elementwise_kernel|reduce_kernel|CatArray|index"),
    ("runtime fill/copy", r"__amd_rocclr"),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
    rows = list(csv.DictReader(open(path)))
    agg = {c: [0.0, 0] for c, _ in CATS}
    agg["other"] = [0.0, 0]
    other = []
    for r in rows:
        n, t, c = r["Name"], float(r["TotalDurationNs"]), int(r["Calls"])
        for cat, pat in CATS:
            if re.search(pat, n):
                agg[cat][0] += t
                agg[cat][1] += c
                break
        else:
            agg["other"][0] += t
            agg["other"][1] += c
            other.append((t, c, n))
    tot = sum(v[0] for v in agg.values())
    ncalls = sum(v[1] for v in agg.values())
    print(f"# {path}: {tot / 1e6 / steps:.2f} ms of kernels, {ncalls / steps:.0f} launches per step ({steps:g} steps)")
    for cat, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{t / 1e6 / steps:8.3f} ms {c / steps:7.1f} launches  {cat}")
    for t, c, n in sorted(other, reverse=True)[:12]:
        print(f"   other: {t / 1e6 / steps:7.3f} ms {c / steps:6.1f}  {n[:110]}")
    if top:
        print("# top kernels")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
            n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])
            n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:100]
            print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms {int(r['Calls']) / steps:6.1f}/step {float(r['AverageNs']) / 1e3:8.1f} us  {n}")


if __name__ == "__main__":
    main()

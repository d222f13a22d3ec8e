"""Copy the files tools/refresh_profiles.sh left under gpurun_out/profiles_rNN/ into profiles/rNN_* and check each one:
a file that is empty, truncated or not what its name says stops the copy.

    python tools/collect_profiles.py r02
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def jsonl(path, min_lines, need=()):
    lines = [l for l in open(path).read().splitlines() if l.strip()]
    assert len(lines) >= min_lines, f"{path}: {len(lines)} lines, expected at least {min_lines}"
    rows = [json.loads(l) for l in lines]
    for r in rows:
        for k in need:
            assert k in r, f"{path}: a line lacks {k!r}"
    return rows


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"profiles_{rnd}")
    dst = os.path.join(ROOT, "profiles")
    assert os.path.isdir(src), f"{src} not found: run tools/refresh_profiles.sh {rnd} on the GPU box first"
    b = jsonl(os.path.join(src, "bench_n1.json"), 1, ("metric", "value", "roofline", "cpu_baseline"))[0]
    assert b["roofline"] and b["cpu_baseline"], "bench line without roofline / cpu_baseline"
    # the lookup kernel's duration in the bench line and in the kernel trace of the same command are one quantity
    k1 = b.get("roofline_hbm")
    assert k1 and k1.get("traffic"), "bench line: roofline_hbm without traffic (counter pass lacks gather_fused_kernel?)"
    roc = [r for r in csv.DictReader(open(os.path.join(src, "bench_kernel_stats.csv"))) if "gather_fused_kernel" in r["Name"]]
    assert roc, "kernel stats lack gather_fused_kernel"
    roc_us = sum(float(r["TotalDurationNs"]) for r in roc) / sum(int(r["Calls"]) for r in roc) / 1e3
    assert abs(roc_us - k1["avg_us"]) <= 0.15 * roc_us, f"K1: bench line {k1['avg_us']} us, rocprofv3 {roc_us:.2f} us"
    print(f"K1 lookup: bench line {k1['avg_us']} us, rocprofv3 {roc_us:.2f} us")
    fd = jsonl(os.path.join(src, "force_dist_bench.json"), 1, ("metric", "value", "roofline", "exchange_check"))[0]
    assert fd["exchange_check"] == "ok" and "row-sharded" in fd["config"]["parallelism"], "force_dist_bench.json is not the row-sharded step"
    fv = jsonl(os.path.join(src, "force_dist_variants.jsonl"), 6, ("metric", "value", "config", "exchange_check"))
    assert all(r["exchange_check"] == "ok" for r in fv)
    assert any("rccl" in r["Name"].lower() for r in csv.DictReader(open(os.path.join(src, "force_dist_kernel_stats.csv")))), \
        "force_dist_kernel_stats.csv holds no RCCL kernel"
    for name in ("force_dist_step_breakdown.txt", "force_dist_two_steps.txt"):
        assert "shard_lookup_kernel" in open(os.path.join(src, name)).read(), name
    v = jsonl(os.path.join(src, "bench_variants.jsonl"), 19, ("metric", "value", "config"))
    kinds = [r["config"]["embedding_update"].split(",")[0] for r in v]
    assert "dense" in kinds and any("lazy" in r["config"]["embedding_update"] for r in v), "variants lack the dense / lazy lines"
    it = jsonl(os.path.join(src, "interact_roofline.jsonl"), 30, ("kernel", "avg_us"))
    names = {r["kernel"] for r in it}
    for k in ("pnn_inner_fwd", "pnn_outer_bwd", "dcn_cross_bwd", "afm_bwd"):
        assert any(n.startswith(k) for n in names), f"interact_roofline.jsonl lacks {k}"
    for name in ("bench_kernel_stats.csv", "interact_kernel_stats.csv"):
        rows = list(csv.DictReader(open(os.path.join(src, name))))
        assert len(rows) >= 10 and "AverageNs" in rows[0], f"{name}: not a rocprofv3 kernel_stats file"
    assert any("gemm_f32_dma_kernel" in r["Name"] for r in csv.DictReader(open(os.path.join(src, "bench_kernel_stats.csv"))))
    assert any("pnn_inner" in r["Name"] for r in csv.DictReader(open(os.path.join(src, "interact_kernel_stats.csv"))))
    sb = open(os.path.join(src, "step_breakdown.txt")).read()
    assert "us/step" in sb and "gemm_f32_dma_kernel" in sb, "step_breakdown.txt is not a step breakdown"
    t = json.load(open(os.path.join(src, "pmc_traffic.json")))
    assert any(k.startswith("gemm_f32_dma_kernel") for k in t["kernels"]) and any("gather" in k for k in t["kernels"])
    mu = json.load(open(os.path.join(src, "mfma_util.json")))
    assert any(k.startswith("gemm_f32") and v["SQ_INSTS_MFMA"] > 0 for k, v in mu["kernels"].items()), "mfma_util.json lacks the GEMM"
    for name in ("bench_n1.json", "bench_variants.jsonl", "interact_roofline.jsonl", "bench_kernel_stats.csv",
                 "interact_kernel_stats.csv", "step_breakdown.txt", "pmc_traffic.json", "mfma_util.json",
                 "force_dist_bench.json", "force_dist_variants.jsonl", "force_dist_kernel_stats.csv",
                 "force_dist_step_breakdown.txt", "force_dist_two_steps.txt"):
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, f"{rnd}_{name}"))
        print("profiles/" + f"{rnd}_{name}")
    # bench.py reads the traffic counters from the newest round's file
    shutil.copyfile(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, "pmc_traffic.json"))


if __name__ == "__main__":
    main()

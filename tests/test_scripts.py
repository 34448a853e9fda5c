"""Host-side tooling around the measurement: the PMC traffic summariser and bench.py's lookup of it."""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_pass(d, counter, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "p_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for disp, kern, val in rows:
            w.writerow({"Dispatch_Id": disp, "Kernel_Name": kern, "Counter_Name": counter,
                        "Counter_Value": val})


def test_pmc_traffic_summary(tmp_path):
    """(2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, 4-wave and split-K launches of one tile shape
    folded into the family bench.py's event brackets use, counters of other kernels kept apart."""
    pt = _load(os.path.join(ROOT, "scripts", "pmc_traffic.py"), "pmc_traffic")
    k4 = "void (anonymous namespace)::fast_conv_kernel<64, 128, false, 4>((anonymous namespace)::FastConvArgs)"
    k8 = "void (anonymous namespace)::fast_conv_sk_kernel<64, 128, true>((anonymous namespace)::FastConvArgs)"
    kb = "(anonymous namespace)::bn_apply_vec_kernel(unsigned short const*, int)"
    fdir, wdir = str(tmp_path / "f"), str(tmp_path / "w")
    _write_pass(fdir, "FETCH_SIZE", [(1, k4, 100.0), (2, k8, 300.0), (3, kb, 50.0)])
    _write_pass(wdir, "WRITE_SIZE", [(1, k4, 10.0), (2, k8, 30.0), (3, kb, 50.0)])
    out = str(tmp_path / "t.json")
    argv = sys.argv
    try:
        sys.argv = ["pmc_traffic.py", fdir, wdir, out]
        pt.main()
    finally:
        sys.argv = argv
    fam = json.load(open(out))["families"]
    conv = fam["fast_conv_kernel<64, 128, *>"]
    assert conv["launches_fetch_pass"] == 2 and conv["launches_write_pass"] == 2
    assert conv["hbm_bytes_per_launch"] == int((2 * 200.0 + 20.0) * 1024)
    assert fam["bn_apply_vec_kernel"]["hbm_bytes_per_launch"] == int((2 * 50.0 + 50.0) * 1024)
    assert pt.family("void (anonymous namespace)::halo_wgrad_kernel<true, false>(x)") == "halo_wgrad_kernel<*>"
    assert pt.family("void (anonymous namespace)::hconv_kernel<128, false, 5, 1>(x)") == "hconv_kernel<128, *>"
    assert pt.family("void (anonymous namespace)::hconv_rw_kernel<true>(x)") == "hconv_kernel<64, *>"
    assert pt.family("void (anonymous namespace)::hwgrad_kernel<true, 5>(x)") == "hwgrad_kernel<*>"
    assert pt.family("void (anonymous namespace)::wstem_fwd_kernel<4, 5>(x)") == "stem_fwd_kernel<*>"


def test_bench_reads_the_committed_traffic_summary():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_module")
    summary = json.load(open(bench.PMC_TRAFFIC_FILE))["workloads"]
    assert os.path.basename(bench.PMC_TRAFFIC_FILE) == "r06_pmc_traffic.json"
    for family in ("hconv_kernel<128, *>", "hwgrad_kernel<*>", "sconv_kernel<*>", "swgrad_kernel<*>"):
        assert bench.pmc_traffic(family) == \
            summary["cifar"]["families"][family]["hbm_bytes_per_launch"] > 0
    for family in ("hconv_kernel<128, *>", "hwgrad_kernel<*>", "hconv_kernel<64, *>"):
        assert bench.pmc_traffic(family, "resnet128_dstep") == \
            summary["resnet128_dstep"]["families"][family]["hbm_bytes_per_launch"] > 0
    for family in ("hconv_kernel<128, *>", "fast_conv_kernel<128, 128, *>", "fast_wgrad_kernel<64, *>"):
        assert bench.pmc_traffic(family, "biggan128_bs256") == \
            summary["biggan128_bs256"]["families"][family]["hbm_bytes_per_launch"] > 0
    assert bench.pmc_traffic("no_such_kernel") is None


def test_gpu_launcher_is_valid_shell_and_documents_its_tasks():
    """scripts/gpu.sh is the one launcher of every GPU visit: it must at least parse, reject unknown
    tasks, and every task of its case statement must be described in scripts/README.md."""
    import re
    import subprocess
    sh = os.path.join(ROOT, "scripts", "gpu.sh")
    assert subprocess.run(["bash", "-n", sh]).returncode == 0
    r = subprocess.run(["bash", sh, "no_such_task"], capture_output=True, text=True,
                       env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
    assert r.returncode == 2 and "unknown task" in r.stdout
    body = open(sh).read()
    tasks = re.findall(r"^  ([a-z]+)\) ", body, flags=re.M)
    assert {"full", "tests", "bench", "launches", "stats", "traffic", "ab", "dp", "final"} <= set(tasks)
    readme = open(os.path.join(ROOT, "scripts", "README.md")).read()
    for t in tasks:
        assert "`%s" % t in readme, t


def test_graph_timeline_on_a_fabricated_trace(tmp_path):
    """scripts/graph_timeline.py (the source of profiles/r05_cifar_graph_timeline.txt): a serial chain
    with one gap and one overlapping pair -- span, sum of durations, union of busy intervals and idle
    time come out as constructed; kernels behind the first calibration launch are ignored."""
    import subprocess
    rows = [("k_a", 1000, 2000), ("k_b", 2000, 3500), ("k_c", 3000, 4000),     # b and c overlap by 500
            ("counter_add_kernel(long*, long)", 6000, 6500),                   # 2000 ns idle before it
            ("calib_mfma_kernel(int, float*)", 9000, 99000), ("k_late", 99000, 99500)]
    path = str(tmp_path / "trace.csv")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", name, s, e])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "graph_timeline.py"), path, "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    head = r.stdout.splitlines()[0]
    assert "kernels 4 " in head and "span 5.5 us" in head and "sum of durations 4.0 us" in head
    assert "union busy 3.5 us" in head and "idle 2.0 us" in head and "(1 gaps" in head
    assert "calib" not in r.stdout and "k_late" not in r.stdout

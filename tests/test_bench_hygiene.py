"""bench.py's line must not carry typed-in measurements: roofline.traffic comes from profiles/traffic.json, which tools/pmc_summary.py --emit writes
out of rocprofv3 --pmc CSVs."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _csv(path, kernel, counter, values):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for i, v in enumerate(values):
            w.writerow([i, kernel, counter, v, 1000 * i, 1000 * i + 500])


def test_pmc_summary_emits_what_bench_reads(tmp_path):
    k1 = "void ccsm::gru_layer12_mx_kernel<false, false, false, false, 3>(HIP_vector_type<unsigned int, 4u> const*)"
    k2 = "void ccsm::gru_layer12_mx_kernel<true, false, false, false, 3>(HIP_vector_type<unsigned int, 4u> const*)"
    root = tmp_path / "pmc"
    _csv(str(root / "fetch" / "a_counter_collection.csv"), k1, "FETCH_SIZE", [2000.0, 2000.0])
    _csv(str(root / "fetch" / "b_counter_collection.csv"), k2, "FETCH_SIZE", [1000.0])
    _csv(str(root / "write" / "a_counter_collection.csv"), k1, "WRITE_SIZE", [600.0])
    _csv(str(root / "write" / "b_counter_collection.csv"), k2, "WRITE_SIZE", [400.0])
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(root), str(tmp_path / "pmc.md"), "t", "--emit", str(out),
                    "--precision", "4", "--sites", "1024", "--source", "test"], check=True, stdout=subprocess.DEVNULL)
    doc = json.load(open(out))
    (ent,) = doc["kernels"]
    assert ent["kernel"] == "gru_layer12_mx_kernel" and ent["precision"] == 4 and ent["launch_variants_averaged"] == 2
    assert ent["fetch_size_KiB"] == 1500.0 and ent["write_size_KiB"] == 500.0
    assert ent["bytes_per_site"] == (2 * 1500.0 + 500.0) * 1024.0 / 1024
    # bench.py reads exactly this structure
    sys.path.insert(0, ROOT)
    import bench
    old = bench.TRAFFIC_FILE
    try:
        bench.TRAFFIC_FILE = str(out)
        per_site, src = bench.traffic_of(4, "gru_layer12_mx_kernel")
        assert per_site == ent["bytes_per_site"] and "test" in src
        assert bench.traffic_of(3, "gru_layer12_f3s_kernel") == (None, None)
    finally:
        bench.TRAFFIC_FILE = old


def test_committed_traffic_file_is_the_rounds_pmc_pass():
    sys.path.insert(0, ROOT)
    import bench
    per_site, src = bench.traffic_of(4, "gru_layer12_mx_kernel")
    assert per_site is not None and 1e5 < per_site < 1e6 and "12288 sites per launch" in src      # the 512-workgroup shape that is timed


def test_no_typed_in_measurement_in_the_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"value": 1.681e6' not in src and "TRAFFIC = {" not in src

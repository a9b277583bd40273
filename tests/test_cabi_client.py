"""The drop-in boundary from compiled code: tests/cabi/cabi_client.c (plain C, gcc) against libccsm.so.
CPU: it builds, links, loads the library and exercises the error path.  GPU: a forward through ccsm_forward_host and
ccsm_submit_host / ccsm_wait_host equals the NumPy oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

LIBDIR = os.path.join(ROOT, "ccsmeth_amd", "lib")


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "libccsm.so")):
        subprocess.check_call(["python", "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    exe = str(tmp_path_factory.mktemp("cabi") / "cabi_client")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "cabi_client.c"),
                           "-L" + LIBDIR, "-lccsm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    return exe


def test_c_client_builds_links_and_reports_errors(client):
    out = subprocess.run([client, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    text = out.stdout.decode()
    assert out.returncode == 0, text
    assert "libccsm" in text and "error path ok" in text


@pytest.mark.gpu
def test_c_client_forward_equals_oracle(client, tmp_path):
    from ccsmeth_amd.utils import synth
    from oracle import attbigru2s_oracle as orc
    n = 70
    w = synth.synth_weights(71)
    s = synth.synth_sites(n, 72)
    h1, h2 = synth.synth_h0(n, 73)
    _, want = orc.attbigru2s_forward(w, s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"], h1, h2)
    path = str(tmp_path / "case.bin")
    f32 = lambda a: np.ascontiguousarray(a, dtype="<f4").tobytes()  # noqa: E731
    with open(path, "wb") as wf:
        wf.write(np.int32(n).tobytes())
        for v in w.values():                                  # state_dict order
            wf.write(f32(v))
        for k in ("1", "2"):
            wf.write(np.ascontiguousarray(s["kmer" + k], dtype=np.uint8).tobytes())
            wf.write(f32(s["ipd" + k])); wf.write(f32(s["pw" + k])); wf.write(f32(s["npass" + k]))
        wf.write(f32(h1)); wf.write(f32(h2)); wf.write(f32(want))
    out = subprocess.run([client, path, "1e-4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0, out.stdout.decode()
    assert "capacity_status 5" in out.stdout.decode()

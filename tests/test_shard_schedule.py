"""CPU test of the chunk / peer schedule of csp_minsnap_solve_batch_sharded's device-memory form (RCCL scatter / solve / gather):
the product's schedule template (cs-pathplan_amd/csrc/minsnap_shard_schedule.h) over a recording in-memory transport
(cs-pathplan_amd/host/shard_schedule_check.cpp).  The RCCL transport itself has never run on more than one GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shard") / "shard_schedule_check")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "cs-pathplan_amd", "host", "shard_schedule_check.cpp"), "-o", out])
    return out


@pytest.mark.parametrize("B,ndev,root,nchunks", [(524288, 8, 0, 4), (65536, 2, 1, 1), (1000, 3, 2, 5), (7, 8, 3, 2), (4096, 1, 0, 3), (65537, 4, 0, 4)])
def test_schedule_over_a_recording_transport(exe, B, ndev, root, nchunks):
    r = subprocess.run([exe, str(B), str(ndev), str(root), str(nchunks)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok")
    # every device solves every non-empty piece once; two groups per chunk (scatter, gather)
    solves = int(r.stdout.split("solves=")[1])
    assert solves <= ndev * nchunks and int(r.stdout.split("groups=")[1].split()[0]) == 2 * nchunks


def test_schedule_rejects_bad_arguments(exe):
    assert subprocess.run([exe, "0", "2", "0", "1"], capture_output=True).returncode == 1
    assert subprocess.run([exe, "10", "2", "2", "1"], capture_output=True).returncode == 1

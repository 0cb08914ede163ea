"""Generates the round-3 fixture tests/golden/F2b_readme_uav31_merged.json -- TEST INFRASTRUCTURE.

Run in the dev container:  python oracle/gen_golden_r3.py

F2b  the README uav31_0 flight AT THE SHAPE getPlan REALLY FEEDS the solver.  preparePlanningWaypoints
     (/root/reference/uavPathPlanning.cpp:2643-2664) drops every midway waypoint i whose 2-D distance to waypoint i+1 is
     <= 200 m ("merging waypoint i to next"); of the README's seven ENU waypoints (readme.md:14-20) the sixth lies 99.96 m
     from the seventh, so the solver sees SIX waypoints = FIVE segments (SURVEY.md section 8c, F2 note), not the six
     segments of fixture F2.  Same orders / speeds / yaml case as F2.
Expected values come from oracle/numpy_ref.py (the independent numpy/LAPACK restatement of
/root/reference/math_util/minimum_snap.cpp:22-649).  PARITY UNPINNED with respect to the real Eigen build: the reference
ships no coefficient goldens and cannot be built here (no Eigen).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import numpy_ref as nr  # noqa: E402
from oracle.gen_golden_r2 import solve_case  # noqa: E402
from tests import synth  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def merge_close_waypoints(P, min_dist=200.0):
    """uavPathPlanning.cpp:2643-2664 with every waypoint a midway point: keep i < last only if hypot2d(p_i, p_{i+1}) > min_dist."""
    keep = [i for i in range(len(P) - 1) if np.hypot(P[i, 0] - P[i + 1, 0], P[i, 1] - P[i + 1, 1]) > min_dist]
    return P[keep + [len(P) - 1]]


def main():
    P = merge_close_waypoints(synth.README_UAV31_ENU)
    assert P.shape == (6, 3) and np.array_equal(P[-1], synth.README_UAV31_ENU[-1]) and np.array_equal(P[-2], synth.README_UAV31_ENU[4])
    cases = []
    for o in (2, 3, 4):
        for V in (200.0, 30.0):
            T = nr.time_allocation(P, V, 1.0)
            cases.append(solve_case("uav31_merged_o%d_V%d" % (o, int(V)), o, P, T,
                                    note="README waypoints after getPlan's 200 m merge (waypoint 5 dropped): 5 segments"))
    T = nr.time_allocation(P, 200.0, 1.0)
    cases.append(solve_case("uav31_merged_yaml", 2, P, T, pw=1e-7, vw=0.01, note="shipped yaml parameters, merged waypoint list"))
    json.dump({"fixture": "F2b", "cases": cases}, open(os.path.join(OUT, "F2b_readme_uav31_merged.json"), "w"), indent=1)
    print("F2b: %d cases, S = %d" % (len(cases), len(P) - 1))


if __name__ == "__main__":
    main()

"""Hit-list comparison used by the parity checks.

TEST INFRASTRUCTURE (like everything under oracle/): imported by tests/ and
__graft_entry__.smoke() only.

Two float32 pipelines with different butterfly orders agree to ~1e-5 of a block's
largest power, so a cell whose power sits within that distance of powcut[stage]
can cross the threshold on one side and not on the other.  Such cells are the only
differences a correct implementation may show; everything else must match by
(stage, z row, column) with powers inside the tolerance of the caller.
"""
import numpy as np

PLANE_TOL = 2e-5          # relative agreement of plane powers between the two FFT pipelines
EDGE_TOL = 4 * PLANE_TOL  # how close to its threshold a one-sided hit must be


def compare_lists(powcut, got_hits, want_hits):
    """Match GPU hits (api.HIT_DTYPE) with oracle hits (binding.HIT_DTYPE) by
    (stage, zind, rind).  Returns (rel, edge_g, edge_w): the relative power
    differences of the matched hits, and for the hits found on one side only
    |power / powcut[stage] - 1| (GPU-only, oracle-only)."""
    powcut = np.asarray(list(powcut), dtype=np.float64)
    key = lambda st, z, r: (st.astype(np.int64) << 56) | (z.astype(np.int64) << 40) | r.astype(np.int64)
    st_w_all = np.log2(want_hits["numharm"]).astype(np.int64)
    kg = key(got_hits["stage"], got_hits["zind"], got_hits["rind"])
    kw = key(st_w_all, want_hits["zind"], want_hits["rind"])
    common, ig, iw = np.intersect1d(kg, kw, return_indices=True)
    rel = np.abs(got_hits["power"][ig] - want_hits["power"][iw]) / want_hits["power"][iw]
    only_g = np.setdiff1d(np.arange(len(kg)), ig)
    only_w = np.setdiff1d(np.arange(len(kw)), iw)
    edge_g = np.abs(got_hits["power"][only_g] / powcut[got_hits["stage"][only_g]] - 1.0)
    edge_w = np.abs(want_hits["power"][only_w] / powcut[st_w_all[only_w]] - 1.0)
    return rel, edge_g, edge_w


def assert_same_search(powcut, got_hits, want_hits, min_matched=0.98, power_tol=10 * PLANE_TOL):
    """The acceptance rule of the end-to-end parity tests, in one place."""
    rel, edge_g, edge_w = compare_lists(powcut, got_hits, want_hits)
    assert len(rel) >= min_matched * max(len(want_hits), 1), (len(rel), len(want_hits))
    if len(rel):
        assert rel.max() <= 1e-3, rel.max()           # the stated bar for powers
        assert rel.max() <= power_tol, rel.max()      # what float32 round-off explains
    assert (edge_g <= EDGE_TOL).all(), ("GPU-only hits away from a threshold", edge_g.max())
    assert (edge_w <= EDGE_TOL).all(), ("oracle-only hits away from a threshold", edge_w.max())
    return rel, edge_g, edge_w

"""Pin the CPU oracle: it must reproduce, bit for bit, every golden vector that was generated
from the reference's own code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import stg_oracle as orc
from tests.util import GAT_SHAPES, GCN_WIDTHS, golden

CSR_KEYS = ("row_offset", "column_indices", "eids")


@pytest.mark.parametrize("tag", ["n1", "n5", "n64", "n2708"])
def test_csr_golden(tag):
    d = golden(f"csr_{tag}.npz")
    n = int(d["num_nodes"])
    g = orc.build_graph(d["src"], d["dst"], n, d["weights_by_eid"])
    for side, c in (("fwd", g.fwd), ("bwd", g.bwd)):
        for k in CSR_KEYS:
            assert np.array_equal(getattr(c, k), d[f"{side}_{k}"]), (side, k)
        # node_ids come from the reference's compiled std::sort (tie order unspecified): the same
        # degree sequence, and a permutation of the vertices
        deg, ref_nid = np.diff(c.row_offset), d[f"{side}_node_ids"]
        assert np.array_equal(deg[c.node_ids], deg[ref_nid]) and np.all(np.diff(deg[ref_nid]) <= 0)
        assert sorted(c.node_ids.tolist()) == sorted(ref_nid.tolist()) == list(range(n))
    assert np.array_equal(g.in_degrees(), d["in_degrees"])
    assert np.array_equal(g.out_degrees(), d["out_degrees"])
    assert np.array_equal(g.fwd.weighted_out_degrees.astype(np.int32), d["weighted_in_degrees"])
    assert np.array_equal(np.stack([d["src"], d["dst"]], 1)[g.perm_fwd], d["sorted_inplace"])
    assert len({(int(a), int(b)) for a, b in zip(d["src"], d["dst"])}) == int(d["num_edges"])


def test_csr_survey_example():
    """The 5-edge CSR recorded from the reference in SURVEY.md 8(c)."""
    g = orc.build_graph([0, 1, 2, 0, 3], [1, 0, 1, 2, 2], 4)
    assert g.fwd.row_offset.tolist() == [0, 1, 3, 5, 5] and g.fwd.column_indices.tolist() == [1, 0, 2, 0, 3]
    assert g.fwd.eids.tolist() == [0, 1, 2, 3, 4]
    assert g.bwd.row_offset.tolist() == [0, 2, 3, 4, 5] and g.bwd.column_indices.tolist() == [1, 2, 0, 1, 2]
    assert g.bwd.eids.tolist() == [1, 3, 0, 2, 4]


@pytest.mark.parametrize("gname", ["static", "naive"])
def test_gcn_golden(gname):
    d = golden("gcn.npz")
    n = int(d["num_nodes"])
    g = orc.build_graph(d["src"], d["dst"], n)
    for side, c in (("fwd", g.fwd), ("bwd", g.bwd)):
        for k in CSR_KEYS:
            assert np.array_equal(getattr(c, k), d[f"{gname}_{side}_{k}"])
    norm = d[f"{gname}_norm"]
    for F in GCN_WIDTHS:
        fa = orc.ref_active_columns(F)
        for use_ew in (False, True):
            tag = f"{gname}_F{F}_{'ew' if use_ew else 'now'}"
            w = d["edge_weight_by_eid"] if use_ew else None
            out = orc.gcn_agg(d[tag + "_x"], norm, norm, g.fwd, ew=w, use_node_ids=gname == "naive", f_active=fa)
            gx = orc.gcn_agg(d[tag + "_R"], norm, norm, g.bwd, ew=w, use_node_ids=gname == "naive", f_active=fa)
            assert np.array_equal(out, d[tag + "_out"]), tag
            assert np.array_equal(gx, d[tag + "_grad_x"]), tag
            if fa < F:          # reference defect D1: the tail columns were never computed
                assert not out[:, fa:].any() and not d[tag + "_out"][:, fa:].any()
            # OpenMP build: same bits regardless of thread count
            assert np.array_equal(orc.gcn_agg(d[tag + "_x"], norm, norm, g.fwd, ew=w, f_active=fa, omp=True),
                                  orc.gcn_agg(d[tag + "_x"], norm, norm, g.fwd, ew=w, f_active=fa))


@pytest.mark.parametrize("H,D", GAT_SHAPES)
def test_gat_golden(H, D):
    d = golden("gat.npz")
    n = int(d["num_nodes"])
    g = orc.build_graph(d["src"], d["dst"], n)
    tag = f"H{H}_D{D}"
    el, er, feat = d[tag + "_k_el"], d[tag + "_k_er"], d[tag + "_k_feat"]
    ha, hda = orc.ref_active_columns(H), orc.ref_active_columns(H * D)
    A, S = orc.gat_k0(el, er, g.fwd, g.num_edges, h_active=ha)
    out = orc.gat_k1(A, S, feat, g.fwd, hd_active=hda)
    gf, gel, ger = orc.gat_bwd(A, S, out, d[tag + "_R"], el, er, feat, g.bwd, hd_active=hda)
    assert np.array_equal(A, d[tag + "_k_A"]) and np.array_equal(S, d[tag + "_k_S"])
    assert np.array_equal(out, d[tag + "_out"])
    assert np.array_equal(gf, d[tag + "_k_grad_feat"])
    assert np.array_equal(gel, d[tag + "_k_grad_el"]) and np.array_equal(ger, d[tag + "_k_grad_er"])
    # reference defect D2: attention is uniform, S equals the in-degree
    assert np.array_equal(S[:, 0, 0], g.in_degrees().astype(np.float32))
    assert S[7].sum() == 0 and not out[7].any()          # the in-degree-0 vertex


# ------------------------------------------------------------------- larger reference-generated graphs (round 2)
def cora_inputs(d, tag, n, shape_tail):
    """x, R of a gcn_cora / gat_cora case re-drawn from the stored seed (numpy PCG64: the same bits on every host)."""
    rng = np.random.default_rng(int(d[tag + "_seed"]))
    return rng, rng.standard_normal((n,) + shape_tail, dtype=np.float32), None


def test_gcn_golden_n200_hub_selfloops_isolated():
    """N = 200 with a hub of in-degree >= 90 (a row longer than a wave), self-loops and isolated vertices, all widths."""
    d = golden("gcn_n200.npz")
    n = int(d["num_nodes"])
    g = orc.build_graph(d["src"], d["dst"], n)
    assert int(d["in_degrees"].max()) >= 90 and (d["in_degrees"] == 0).sum() >= 2 and (d["src"] == d["dst"]).sum() >= 3
    for side, c in (("fwd", g.fwd), ("bwd", g.bwd)):
        for k in CSR_KEYS:
            assert np.array_equal(getattr(c, k), d[f"{side}_{k}"])
    norm = d["norm"]
    for F in GCN_WIDTHS:
        fa = orc.ref_active_columns(F)
        for use_ew in (False, True):
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            w = d["edge_weight_by_eid"] if use_ew else None
            assert np.array_equal(orc.gcn_agg(d[tag + "_x"], norm, norm, g.fwd, ew=w, f_active=fa), d[tag + "_out"]), tag
            assert np.array_equal(orc.gcn_agg(d[tag + "_R"], norm, norm, g.bwd, ew=w, f_active=fa), d[tag + "_grad_x"]), tag


def test_gcn_golden_cora_shaped():
    """The benchmark's Cora-shaped graph (N = 2708, E = 10556, max in-degree 168): sampled rows in full + fp64 column
    sums over all rows (a bit-exact result reproduces them exactly)."""
    d = golden("gcn_cora.npz")
    n = int(d["num_nodes"])
    g = orc.build_graph(d["src"], d["dst"], n)
    assert int(d["max_in_degree"]) >= 150                 # the hub (expected degree capped at 168: 159 drawn)
    for side, c in (("fwd", g.fwd), ("bwd", g.bwd)):
        for k in CSR_KEYS:
            assert np.array_equal(getattr(c, k), d[f"{side}_{k}"])
    norm, rows = d["norm"], d["rows"]
    for F in (7, 16, 64, 300):
        fa = orc.ref_active_columns(F)
        for use_ew in (False, True):
            tag = f"F{F}_{'ew' if use_ew else 'now'}"
            rng = np.random.default_rng(int(d[tag + "_seed"]))
            x = rng.standard_normal((n, F), dtype=np.float32)
            R = rng.standard_normal((n, F), dtype=np.float32)
            assert np.array_equal(x[rows], d[tag + "_x_rows"]) and np.array_equal(R[rows], d[tag + "_R_rows"])
            w = d["edge_weight_by_eid"] if use_ew else None
            out = orc.gcn_agg(x, norm, norm, g.fwd, ew=w, f_active=fa)
            gx = orc.gcn_agg(R, norm, norm, g.bwd, ew=w, f_active=fa)
            assert np.array_equal(out[rows], d[tag + "_out_rows"]) and np.array_equal(gx[rows], d[tag + "_grad_x_rows"]), tag
            assert np.array_equal(out.astype(np.float64).sum(0), d[tag + "_out_colsum"]), tag
            assert np.array_equal(gx.astype(np.float64).sum(0), d[tag + "_grad_x_colsum"]), tag


@pytest.mark.parametrize("H,D", [(8, 8), (8, 64)])
def test_gat_golden_cora_shaped(H, D):
    d = golden("gat_cora.npz")
    n = int(d["num_nodes"])
    g = orc.build_graph(d["src"], d["dst"], n)
    tag, rows = f"H{H}_D{D}", d["rows"]
    feat = (d[tag + "_x"].astype(np.float64) @ d[tag + "_fc_weight"].astype(np.float64).T).astype(np.float32).reshape(n, H, D)
    assert np.array_equal(feat[rows], d[tag + "_k_feat_rows"])        # exact by construction (dyadic inputs)
    rng = np.random.default_rng(int(d[tag + "_seed"]))
    rng.integers(-8, 9, (n, 6)), rng.integers(-16, 17, (H * D, 6))     # the generator's draws before R
    R = rng.standard_normal((n, H, D), dtype=np.float32)
    assert np.array_equal(R[rows], d[tag + "_R_rows"])
    el, er = d[tag + "_k_el"], d[tag + "_k_er"]
    A, S = orc.gat_k0(el, er, g.fwd, g.num_edges)
    out = orc.gat_k1(A, S, feat, g.fwd)
    gf, gel, ger = orc.gat_bwd(A, S, out, R, el, er, feat, g.bwd)
    assert np.array_equal(A, d[tag + "_k_A"]) and np.array_equal(S, d[tag + "_k_S"])
    assert np.array_equal(out[rows], d[tag + "_out_rows"])
    assert np.array_equal(out.astype(np.float64).sum(0), d[tag + "_out_colsum"])
    assert np.array_equal(gf[rows], d[tag + "_k_grad_feat_rows"])
    assert np.array_equal(gf.astype(np.float64).sum(0), d[tag + "_k_grad_feat_colsum"])
    assert np.array_equal(gel, d[tag + "_k_grad_el"]) and np.array_equal(ger, d[tag + "_k_grad_er"])

"""GPU parity tests (-m gpu): libpbc_hip.so through its C-ABI vs the reference's golden
vectors and vs the CPU oracle on seeded inputs -- bit-exact on element_to_bytes output."""
import numpy as np
import pytest

import oracle
from conftest import G2_HASH, G2_COMPRESS, G2_XONLY, XONLY, check_x_only, check_x_only_g2, golden, OTHER, GENERIC_A, GENERIC_OTHER, GENERIC_F, FILES_OF, param_value, key_of

pytestmark = pytest.mark.gpu

Q_A = 8780710799663312522437781984754049815806883199414208211028653399266475630880222957078625179422662221423155858769582317459277713367317481324925129998224791
R_A = 730750818665451621361119245571504901405976559617


def _be(x, n):
    return np.frombuffer(int(x).to_bytes(n, "big"), np.uint8)


def test_fq_ops_vs_oracle(hip_a, oracle_a):
    """guru/checkfp.c pattern: same random inputs through both backends, compare bytes op by op."""
    rng = np.random.default_rng(7)
    n = 1000
    xs = [int.from_bytes(rng.bytes(64), "big") % Q_A for _ in range(n - 4)] + [0, 1, Q_A - 1, Q_A - 2]
    ys = [int.from_bytes(rng.bytes(64), "big") % Q_A for _ in range(n - 4)] + [Q_A - 1, 0, Q_A - 1, 2]
    A = np.stack([_be(x, 64) for x in xs])
    B = np.stack([_be(y, 64) for y in ys])
    for op in range(7):
        got = hip_a.fq_op(op, A, B)
        want = oracle_a.fq_op(op, A, B)
        if op == 3:                    # invert: "requires nonzero" (montfp.c:399); skip the 0 input
            keep = np.array([x != 0 for x in xs])
            got, want = got[keep], want[keep]
        assert np.array_equal(got, want), "fq op %d" % op


def test_fq_from_bytes_reduces_mod_q(hip_a, oracle_a):
    """fp_from_bytes accepts values >= q and reduces them (montfp.c:498-517)."""
    A = np.stack([_be(Q_A + 5, 64), _be(2**512 - 1, 64), _be(Q_A, 64)])
    B = np.stack([_be(1, 64)] * 3)
    assert np.array_equal(hip_a.fq_op(0, A, B), oracle_a.fq_op(0, A, B))


def test_kat(hip_a):
    """pbc/pairing_test.pbc:3-10 through the GPU."""
    v = golden("a_kat.vec")
    assert np.array_equal(hip_a.element_pairing(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("name", ["a_rand32.vec", "a_edge20.vec", "a_chain1024.vec"])
def test_pairing_matches_reference_vectors(hip_a, name):
    v = golden(name)
    assert np.array_equal(hip_a.element_pairing(v.g1, v.g2), v.gt)


def test_pairing_vs_oracle_cross_pairs(hip_a, oracle_a):
    """Inputs the fixtures do not contain: all (P_i, Q_j) cross pairs of a seeded subset."""
    v = golden("a_chain1024.vec")
    rng = np.random.default_rng(3)
    i = rng.integers(0, 1024, 300)
    j = rng.integers(0, 1024, 300)
    got = hip_a.element_pairing(v.g1[i], v.g2[j])
    want = oracle_a.pairing_batch(v.g1[i], v.g2[j])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 127, 129, 257])
def test_ragged_batch_sizes(hip_a, n):
    v = golden("a_chain1024.vec")
    out = hip_a.element_pairing(v.g1[:n], v.g2[:n])
    assert out.shape == (n, 128)
    assert np.array_equal(out, v.gt[:n])


def test_identity_inputs_give_one(hip_a):
    """Off-curve bytes -> O (ecc/curve.c:618-621) -> pairing = 1 (pbc_pairing.h:123-130)."""
    v = golden("a_rand32.vec")
    g1, g2 = v.g1[:8].copy(), v.g2[:8].copy()
    g1[0, 127] ^= 1
    g2[1, 127] ^= 1
    g1[2, 10] ^= 0x80
    g1[3] = 0xFF                          # x,y >= q: reduced mod q first, then off-curve
    out = hip_a.element_pairing(g1, g2)
    one = np.zeros(128, np.uint8)
    one[63] = 1
    for k in range(4):
        assert np.array_equal(out[k], one), k
    assert np.array_equal(out[4:], v.gt[4:8])


def test_bilinearity_on_gpu(hip_a, oracle_a):
    """pbc/bilinear.test:24-33: e(aP,bQ) = e(P,Q)^(ab), pairings on the GPU."""
    v = golden("a_rand32.vec")
    a, b = 42, 17 * 2**100 + 101
    P, Q = v.g1[:4], v.g2[:4]
    aP = oracle_a.g_mul(1, P, np.tile(_be(a, 20), (4, 1)))
    bQ = oracle_a.g_mul(2, Q, np.tile(_be(b, 20), (4, 1)))
    lhs = hip_a.element_pairing(aP, bQ)
    base = hip_a.element_pairing(P, Q)
    rhs = oracle_a.gt_pow(base, np.tile(_be(a * b % R_A, 20), (4, 1)))
    assert np.array_equal(lhs, rhs)


def test_device_pointer_api_and_full_size_batch(hip_a, oracle_a):
    """BASELINE config 2 size: 2^20 Type-A pairings in one launch on device-resident
    buffers (all (P_i, Q_j), i,j < 1024).  Size-independent checks on every output:
    e(P_i,Q_j) = e(P0,Q0)^((i+1)(j+1)) = e(P_j,Q_i)  ->  the 1024x1024 result matrix must be
    symmetric and its diagonal must equal the reference's fixture; plus a seeded sample
    against the oracle."""
    import torch
    v = golden("a_chain1024.vec")
    D = 1024
    g1 = torch.from_numpy(v.g1).cuda()
    g2 = torch.from_numpy(v.g2).cuda()
    G1 = g1[:, None, :].expand(D, D, 128).reshape(D * D, 128).contiguous()
    G2 = g2[None, :, :].expand(D, D, 128).reshape(D * D, 128).contiguous()
    GT = torch.empty(D * D, 128, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hip_a.element_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), D * D, s)
    torch.cuda.synchronize()
    M = GT.reshape(D, D, 128)
    assert torch.equal(M, M.transpose(0, 1)), "e(P_i,Q_j) != e(P_j,Q_i) somewhere"
    diag = M[torch.arange(D), torch.arange(D)].cpu().numpy()
    assert np.array_equal(diag, v.gt)
    rng = np.random.default_rng(11)
    ii, jj = rng.integers(0, D, 64), rng.integers(0, D, 64)
    want = oracle_a.pairing_batch(v.g1[ii], v.g2[jj])
    got = M[torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda()].cpu().numpy()
    assert np.array_equal(got, want)


# ---- element_prod_pairing (a_pairings_affine, ecc/a_param.c:1283-1383) -------------------
@pytest.mark.parametrize("name", ["a_prod16x4.vec", "a_prod2x8.vec", "a_prod3x10_edge.vec"])
def test_prod_pairing_matches_reference_vectors(hip_a, name):
    v = golden(name)
    assert np.array_equal(hip_a.element_prod_pairing(v.g1, v.g2, v.k), v.gt)


def test_prod_pairing_vs_oracle_and_naive_product(hip_a, oracle_a):
    """benchmark/multipairing.c:49 / guru/prodpairing_test.c: prod == naive product of pairings."""
    v = golden("a_chain1024.vec")
    k, n = 5, 37
    rng = np.random.default_rng(5)
    i = rng.integers(0, 1024, n * k)
    j = rng.integers(0, 1024, n * k)
    got = hip_a.element_prod_pairing(v.g1[i], v.g2[j], k)
    assert np.array_equal(got, oracle_a.prod_pairing_batch(v.g1[i], v.g2[j], k))
    singles = hip_a.element_pairing(v.g1[i], v.g2[j]).reshape(n, k, 128)
    acc = singles[:, 0]
    for t in range(1, k):
        acc = oracle_a.gt_mul(acc, singles[:, t])
    assert np.array_equal(got, acc)


def test_prod_pairing_shared_squaring_kernel(hip_a, oracle_a):
    """Default: one term per lane (al_miller_kernel + al_prod_finish_kernel).  "hip_prod_shared 1" selects the kernel with one
    product per lane and the squaring of the accumulator shared between its terms (a_pairings_affine's shape): same bytes."""
    import pbc_amd
    from conftest import _param
    P = pbc_amd.Pairing(_param("a") + "hip_prod_shared 1\n")
    for name in ("a_prod16x4.vec", "a_prod2x8.vec", "a_prod3x10_edge.vec", "a_prodfull3x4.vec"):
        v = golden(name)
        assert np.array_equal(P.element_prod_pairing(v.g1, v.g2, v.k), v.gt), name
    v = golden("a_chain1024.vec")
    rng = np.random.default_rng(9)
    for k, n in ((3, 300), (16, 129), (7, 1)):
        i, j = rng.integers(0, 1024, n * k), rng.integers(0, 1024, n * k)
        got = P.element_prod_pairing(v.g1[i], v.g2[j], k)
        assert np.array_equal(got, hip_a.element_prod_pairing(v.g1[i], v.g2[j], k))
        assert np.array_equal(got[:8], oracle_a.prod_pairing_batch(v.g1[i[:8 * k]], v.g2[j[:8 * k]], k))
    P.clear()


def test_prod_pairing_term_lanes_in_several_launches(hip_a):
    """Long product batches are cut into launches of at most 2^22 terms (one workspace record per term);
    "hip_prod_chunk N" shrinks that bound: ragged last chunk, k larger than the chunk, k = 2."""
    import pbc_amd
    from conftest import _param
    v = golden("a_chain1024.vec")
    rng = np.random.default_rng(10)
    for chunk, k, n in ((1000, 16, 200), (5, 7, 9), (300, 2, 777)):
        P = pbc_amd.Pairing(_param("a") + "hip_prod_chunk %d\n" % chunk)
        i, j = rng.integers(0, 1024, n * k), rng.integers(0, 1024, n * k)
        assert np.array_equal(P.element_prod_pairing(v.g1[i], v.g2[j], k), hip_a.element_prod_pairing(v.g1[i], v.g2[j], k))
        P.clear()


def test_prod_pairing_k1_equals_pairing(hip_a):
    v = golden("a_rand32.vec")
    assert np.array_equal(hip_a.element_prod_pairing(v.g1, v.g2, 1), v.gt)


# ---- Types D (d159, MNT k=6) and F (BN k=12): SURVEY.md 8a rows a15-a17 ------------------
DF_FILES = {
    "d": dict(single=["d_rand32.vec", "d_edge20.vec", "d_chain256.vec"], prod=["d_prod16x4.vec", "d_prod3x10_edge.vec"],
              chain="d_chain256.vec", rand="d_rand32.vec"),
    "f": dict(single=["f_rand16.vec", "f_edge10.vec", "f_chain128.vec"], prod=["f_prod4x3.vec", "f_prod3x5_edge.vec"],
              chain="f_chain128.vec", rand="f_rand16.vec"),
}


def _types_built(hips):
    return [t for t in "df" if t in hips]


@pytest.mark.parametrize("t", ["d", "f"])
def test_df_fq_ops_vs_oracle(hips, oracles, t):
    q = {"d": 625852803282871856053922297323874661378036491717,
         "f": 205523667896953300194896352429254920972540065223}[t]
    rng = np.random.default_rng(9)
    xs = [int.from_bytes(rng.bytes(20), "big") % q for _ in range(500)] + [0, 1, q - 1, 2**160 - 1]
    ys = [int.from_bytes(rng.bytes(20), "big") % q for _ in range(500)] + [q - 1, 0, q - 1, 2**160 - 1]
    A = np.stack([_be(x, 20) for x in xs])
    B = np.stack([_be(y, 20) for y in ys])
    for op in range(7):
        got = hips[t].fq_op(op, A, B)
        want = oracles[t].fq_op(op, A, B)
        if op == 3:
            keep = np.array([x % q != 0 for x in xs])
            got, want = got[keep], want[keep]
        assert np.array_equal(got, want), "fq op %d" % op


@pytest.mark.parametrize("t,name", [(t, n) for t in "df" for n in DF_FILES[t]["single"]])
def test_df_pairing_matches_reference_vectors(hips, t, name):
    v = golden(name)
    assert np.array_equal(hips[t].element_pairing(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("t,name", [(t, n) for t in "df" for n in DF_FILES[t]["prod"]])
def test_df_prod_pairing_matches_reference_vectors(hips, t, name):
    v = golden(name)
    assert np.array_equal(hips[t].element_prod_pairing(v.g1, v.g2, v.k), v.gt)


@pytest.mark.parametrize("t", ["a", "d"])
def test_products_over_many_workgroups_use_their_own_workspace_records(hips, t):
    """The product kernels of types a and d keep the per-term Miller state in a global workspace, one record per
    (workgroup, term, lane), sized by the host from the kernel's own record length: 1500 units of 16 terms (12
    workgroups) must all reproduce the reference's four."""
    v = golden("a_prod16x4.vec" if t == "a" else "d_prod16x4.vec")
    reps = 375
    g1, g2 = np.tile(v.g1, (reps, 1)), np.tile(v.g2, (reps, 1))        # records of unit u: rows u k .. u k + k - 1
    got = hips[t].element_prod_pairing(g1, g2, v.k)
    assert np.array_equal(got, np.tile(v.gt, (reps, 1)))


@pytest.mark.parametrize("t", ["d", "f"])
def test_df_cross_pairs_vs_oracle_and_symmetry(hips, oracles, t):
    """(P_i, Q_j) pairs the fixtures do not hold; e(P_i,Q_j) = e(P_j,Q_i) since P_i=(i+1)P0, Q_j=(j+1)Q0."""
    v = golden(DF_FILES[t]["chain"])
    m = 24
    i, j = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    got = hips[t].element_pairing(v.g1[i.ravel()], v.g2[j.ravel()]).reshape(m, m, -1)
    assert np.array_equal(got, got.transpose(1, 0, 2))
    assert np.array_equal(got[np.arange(m), np.arange(m)], v.gt[:m])
    rng = np.random.default_rng(13)
    ii, jj = rng.integers(0, m, 12), rng.integers(0, m, 12)
    assert np.array_equal(got[ii, jj], oracles[t].pairing_batch(v.g1[ii], v.g2[jj]))


@pytest.mark.parametrize("t", ["d", "f"])
@pytest.mark.parametrize("n", [0, 1, 65, 130])
def test_df_ragged_batch_sizes(hips, t, n):
    v = golden(DF_FILES[t]["chain"])
    n = min(n, v.n)
    out = hips[t].element_pairing(v.g1[:n], v.g2[:n])
    assert out.shape == (n, hips[t].length_in_bytes_GT)
    assert np.array_equal(out, v.gt[:n])


@pytest.mark.parametrize("t", ["d", "f"])
def test_df_bilinearity(hips, oracles, t):
    v = golden(DF_FILES[t]["rand"])
    a = 3 * 2**90 + 12345
    P, Q = v.g1[:3], v.g2[:3]
    aP = oracles[t].g_mul(1, P, np.tile(_be(a, 20), (3, 1)))
    lhs = hips[t].element_pairing(aP, Q)
    rhs = oracles[t].gt_pow(hips[t].element_pairing(P, Q), np.tile(_be(a, 20), (3, 1)))
    assert np.array_equal(lhs, rhs)


# ---- the drop-in itself: unmodified reference PBC + integration/pbc_hip_glue.c ------------
@pytest.mark.parametrize("pname", ["a", "d159", "f", "d201", "g149", "e", "a1"])
def test_reference_call_sites_run_on_gpu_through_the_glue(pname):
    """oracle/_ref/glue_test = the reference library (compiled from /root/reference by
    oracle/Makefile) linked with integration/pbc_hip_glue.c: element_pairing(),
    element_prod_pairing() and the new *_batch() calls go through pairing->map /
    pairing->prod_pairings into libpbc_hip.so and are compared with the reference's own CPU
    results via element_cmp."""
    import os
    import subprocess
    import pbc_amd
    if not os.path.exists(oracle.GLUE_TEST):
        pytest.skip("oracle/_ref/glue_test not built (needs /root/reference at build time)")
    env = dict(os.environ, PBC_HIP_LIB=pbc_amd.LIB_PATH)
    n = "40" if pname in ("e", "a1") else "120"             # the CPU side of the comparison is slow for 1 kbit fields
    r = subprocess.run([oracle.GLUE_TEST, os.path.join(pbc_amd.PARAM_DIR, pname + ".param"), n],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


def test_f_generic_hard_part_on_gpu():
    """Non-BN Type-F parameters use plain square-and-multiply over the whole exponent; force it on f.param."""
    import pbc_amd
    from conftest import _param
    v = golden("f_rand16.vec")
    P = pbc_amd.Pairing(_param("f") + "hip_no_bn 1\n")
    assert np.array_equal(P.element_pairing(v.g1, v.g2), v.gt)


def test_f_reference_basis_path_on_gpu():
    """f.param has q = 3 mod 4: its pairing kernels run in the i-basis of F_q^2 (pairing_f.cuh init_stage3: beta -> -1,
    Q mapped on the way in, GT on the way out).  "hip_no_bm1 1" keeps the parameter file's beta -- the path every q = 1 mod 4
    parameter set takes (f_200, f_256 below): both must give the reference's bytes, singles and products."""
    import pbc_amd
    from conftest import _param
    v, w = golden("f_rand16.vec"), golden("f_prod3x5_edge.vec")
    # ... with plain squarings in the hard part, with the parameter file's xi instead of the sparse one (init_stage4),
    # with the word-form steps on E(F_q): every combination of switches is a different instruction stream, all must agree
    for extra in ("hip_no_bm1 1\n", "", "hip_no_cyc 1\n", "hip_no_bm1 1\nhip_no_cyc 1\n", "hip_no_xs 1\n", "hip_no_limb 1\n",
                  "hip_no_xs 1\nhip_no_limb 1\n", "hip_no_cyc 1\nhip_no_limb 1\n"):
        P = pbc_amd.Pairing(_param("f") + extra)
        assert np.array_equal(P.element_pairing(v.g1, v.g2), v.gt)
        assert np.array_equal(P.element_prod_pairing(w.g1, w.g2, w.k), w.gt)
        P.clear()


def test_d_word_form_point_arithmetic_on_gpu():
    """Type d parameters whose q leaves its top 29-bit limb nearly empty run the word-form step routines inside the
    d159 kernels (DConst::limb_ok); force that path on d159.param: singles, products, preprocessing."""
    import pbc_amd
    from conftest import _param
    P = pbc_amd.Pairing(_param("d159") + "hip_no_limb 1\n")
    v = golden("d_rand32.vec")
    assert np.array_equal(P.element_pairing(v.g1, v.g2), v.gt)
    w = golden("d_prod3x10_edge.vec")
    assert np.array_equal(P.element_prod_pairing(w.g1, w.g2, w.k), w.gt)
    w = golden("d_prod16x4.vec")
    assert np.array_equal(P.element_prod_pairing(w.g1, w.g2, w.k), w.gt)
    pp = P.pp_init(v.g1[0])
    assert np.array_equal(pp.apply(v.g2[:8]), P.element_pairing(np.repeat(v.g1[:1], 8, axis=0), v.g2[:8]))
    assert np.array_equal(pp.apply(v.g2[:1]), v.gt[:1])


@pytest.mark.parametrize("t,log2n", [("d", 18), ("f", 18)])
def test_df_full_size_batch_properties(hips, t, log2n):
    """BASELINE configs 3/4: 2^18 pairings in one launch on device-resident buffers.  All
    (P_i, Q_j) of the chain fixture, tiled: e(P_i,Q_j) = e(P0,Q0)^((i+1)(j+1)) = e(P_j,Q_i), so the
    DxD result matrix is symmetric, every tile repeats it, and its diagonal is the fixture."""
    import torch
    v = golden(DF_FILES[t]["chain"])
    P = hips[t]
    D = v.n
    n = 1 << log2n
    L1, L2, LT = P.length_in_bytes_G1, P.length_in_bytes_G2, P.length_in_bytes_GT
    g1 = torch.from_numpy(v.g1).cuda()
    g2 = torch.from_numpy(v.g2).cuda()
    idx = torch.arange(n, device="cuda")
    G1 = g1[(idx // D) % D].contiguous()
    G2 = g2[idx % D].contiguous()
    GT = torch.empty(n, LT, dtype=torch.uint8, device="cuda")
    P.element_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    M = GT.reshape(n // (D * D), D, D, LT)
    assert torch.equal(M[0], M[0].transpose(0, 1))
    for rep in range(1, M.shape[0]):
        assert torch.equal(M[rep], M[0])
    assert np.array_equal(M[0][torch.arange(D), torch.arange(D)].cpu().numpy(), v.gt)


@pytest.mark.parametrize("extra", [0, 1, 77, 128, 300])
def test_f_resident_grid_ragged_sizes(hips, extra):
    """The type f kernel is launched with resident workgroups that stride over the batch (pbc_hip.hip resident_grid):
    batches just above one chip residency -- 1024 workgroups of 128 lanes on an MI355X -- with ragged tails: every unit
    bit-exact (the chain fixture tiled), none left out, none written twice past the end."""
    import torch
    v = golden("f_chain128.vec")
    P = hips["f"]
    n = 1024 * 128 + extra
    g1 = torch.from_numpy(np.tile(v.g1, (-(-n // v.n), 1))[:n].copy()).cuda()
    g2 = torch.from_numpy(np.tile(v.g2, (-(-n // v.n), 1))[:n].copy()).cuda()
    GT = torch.full((n + 64, v.lenT), 0xA5, dtype=torch.uint8, device="cuda")
    P.element_pairing_dev(GT.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = GT.cpu().numpy()
    want = np.tile(v.gt, (-(-n // v.n), 1))[:n]
    assert np.array_equal(got[:n], want)
    assert (got[n:] == 0xA5).all()


def test_a_prod16_full_size_batch_properties(hip_a, oracle_a):
    """BASELINE config 5: 2^18 products of 16 Type-A pairings in one launch.  Size-independent
    checks: reversing the 16 terms of every product must not change a single output byte, and a
    seeded sample equals the oracle's a_pairings_affine."""
    import torch
    v = golden("a_chain1024.vec")
    k, n, D = 16, 1 << 18, 1024
    g1 = torch.from_numpy(v.g1).cuda()
    g2 = torch.from_numpy(v.g2).cuda()
    t = torch.arange(n * k, device="cuda")
    i1, i2 = (t * 7 + t // D) % D, (t * 13 + 5) % D
    G1, G2 = g1[i1].contiguous(), g2[i2].contiguous()
    rev = (torch.arange(n, device="cuda")[:, None] * k + torch.arange(k - 1, -1, -1, device="cuda")[None, :]).reshape(-1)
    R1, R2 = G1[rev].contiguous(), G2[rev].contiguous()
    A = torch.empty(n, 128, dtype=torch.uint8, device="cuda")
    B = torch.empty(n, 128, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hip_a.element_prod_pairing_dev(A.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, k, s)
    hip_a.element_prod_pairing_dev(B.data_ptr(), R1.data_ptr(), R2.data_ptr(), n, k, s)
    torch.cuda.synchronize()
    assert torch.equal(A, B)
    rng = np.random.default_rng(17)
    us = rng.integers(0, n, 6)
    for u in us:
        sl = slice(int(u) * k, int(u) * k + k)
        want = oracle_a.prod_pairing_batch(G1[sl].cpu().numpy(), G2[sl].cpu().numpy(), k)
        assert np.array_equal(A[int(u)].cpu().numpy(), want[0])


# ---- preprocessed pairings: pairing_pp_init / pairing_pp_apply ------
def test_pairing_pp_matches_element_pairing(hip_a, oracle_a):
    v = golden("a_chain1024.vec")
    for pi in (0, 7):
        pp = hip_a.pp_init(v.g1[pi])
        Q = v.g2[:300].copy()
        Q[5, 127] ^= 1                                    # identity second argument
        got = pp.apply(Q)
        want = hip_a.element_pairing(np.tile(v.g1[pi], (300, 1)), Q)
        assert np.array_equal(got, want)
        assert np.array_equal(got[pi], v.gt[pi])           # e(P_i, Q_i) from the reference fixture
        assert np.array_equal(got[:8], oracle_a.pairing_batch(np.tile(v.g1[pi], (8, 1)), Q[:8]))
        pp.clear()
    bad = v.g1[3].copy()
    bad[1] ^= 8
    pp = hip_a.pp_init(bad)                                # first argument deserialises to O
    one = np.zeros(128, np.uint8)
    one[63] = 1
    assert np.array_equal(pp.apply(v.g2[:3]), np.tile(one, (3, 1)))


@pytest.mark.parametrize("t,name", [("d", "d_chain256.vec"), ("d201", "d201_rand12.vec"), ("d278027-190-181", "d278027-190-181_rand12.vec"),
                                    ("g149", "g149_chain64.vec"), ("a1", "a1_chain8.vec")])
def test_pairing_pp_types_d_g(hips, oracles, t, name):
    """d_pairing_pp_init/apply (d_param.c:794-966), g_pairing_pp_init/apply (g_param.c:619-787)."""
    v = golden(name)
    H = hips[t]
    m = min(v.n, 140)                                      # more than one block for the chain fixtures
    for pi in (0, 3):
        pp = H.pp_init(v.g1[pi])
        Q = v.g2[:m].copy()
        Q[5, -1] ^= 1                                      # identity second argument
        got = pp.apply(Q)
        assert np.array_equal(got, H.element_pairing(np.tile(v.g1[pi], (m, 1)), Q))
        assert np.array_equal(got[pi], v.gt[pi])           # e(P_i, Q_i) from the reference fixture
        c = 2 if t == "a1" else 6                          # the oracle needs 0.4 s per 1033-bit pairing
        assert np.array_equal(got[:c], oracles[t].pairing_batch(np.tile(v.g1[pi], (c, 1)), Q[:c]))
        pp.clear()
    bad = v.g1[2].copy()
    bad[1] ^= 8
    one = np.zeros(H.length_in_bytes_GT, np.uint8)
    one[H.length_in_bytes_G1 // 2 - 1] = 1
    assert np.array_equal(H.pp_init(bad).apply(v.g2[:3]), np.tile(one, (3, 1)))


# ---- the other shipped type d parameter files: 175..224-bit q, 6 / 7 word fields, 22..28-byte
# ---- coordinates (not whole words for d277699-175-167, d105171-196-185, d201), and type g
# ---- (g149.param: k = 10, F_q^5 / F_q^10, 19-byte coordinates) --------------------------------
@pytest.mark.parametrize("d", OTHER)
def test_other_type_d_and_g_params_match_reference_vectors(hips, d):
    H = hips[d]
    if d == "a1":
        fb = (param_value(d, "p").bit_length() + 7) // 8
        assert (H.length_in_bytes_G1, H.length_in_bytes_G2, H.length_in_bytes_GT) == (2 * fb, 2 * fb, 2 * fb)
    elif d == "e":
        fb = (param_value(d, "q").bit_length() + 7) // 8
        assert (H.length_in_bytes_G1, H.length_in_bytes_G2, H.length_in_bytes_GT) == (2 * fb, 2 * fb, fb)
    else:
        fb = (param_value(d, "q").bit_length() + 7) // 8
        deg = param_value(d, "k") // 2
        assert (H.length_in_bytes_G1, H.length_in_bytes_G2, H.length_in_bytes_GT) == (2 * fb, 2 * deg * fb, 2 * deg * fb)
    for name in FILES_OF[d][:2]:
        v = golden(name)
        assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt), name
    v = golden(FILES_OF[d][2])
    assert np.array_equal(H.element_prod_pairing(v.g1, v.g2, v.k), v.gt)


@pytest.mark.parametrize("key,name", [("g149", "g149_chain64.vec"), ("e", "e_chain8.vec"), ("a1", "a1_chain8.vec"),
                                      ("d224", "d224_rand12.vec")])
@pytest.mark.parametrize("n", [0, 1, 65, 130])
def test_other_families_ragged_batch_sizes(hips, key, name, n):
    """whole blocks plus a tail, one lane, and the empty batch through the host-buffer path"""
    v = golden(name)
    idx = np.arange(n) % v.n
    out = hips[key].element_pairing(v.g1[idx], v.g2[idx])
    assert out.shape == (n, hips[key].length_in_bytes_GT)
    assert np.array_equal(out, v.gt[idx])


@pytest.mark.parametrize("key,name,log2n", [("g149", "g149_chain64.vec", 16), ("e", "e_chain8.vec", 14), ("a1", "a1_chain8.vec", 13),
                                            ("e", "e_chain8.vec", 17), ("a1", "a1_chain8.vec", 17)])   # 2^17: two waves on every SIMD (round 5)
def test_other_families_device_batch_properties(hips, key, name, log2n):
    """device-pointer API on a batch that fills the chip: all (P_i, Q_j) of the chain fixture, tiled.
    e(P_i, Q_j) = e(P_0, Q_0)^((i+1)(j+1)), so the D x D result matrix is symmetric, every tile repeats
    it and its diagonal is the reference fixture."""
    import torch
    v = golden(name)
    P = hips[key]
    D = v.n
    n = 1 << log2n
    LT = P.length_in_bytes_GT
    g1 = torch.from_numpy(v.g1).cuda()
    g2 = torch.from_numpy(v.g2).cuda()
    idx = torch.arange(n, device="cuda")
    G1 = g1[(idx // D) % D].contiguous()
    G2 = g2[idx % D].contiguous()
    GT = torch.empty(n, LT, dtype=torch.uint8, device="cuda")
    P.element_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    M = GT.reshape(n // (D * D), D, D, LT)
    assert torch.equal(M[0], M[0].transpose(0, 1))
    for rep in range(1, M.shape[0]):
        assert torch.equal(M[rep], M[0])
    assert np.array_equal(M[0][torch.arange(D), torch.arange(D)].cpu().numpy(), v.gt)


@pytest.mark.parametrize("key,name", [("a", "a_rand32.vec"), ("a1", "a1_rand6.vec"), ("e", "e_rand6.vec")])
def test_symmetric_types_same_point_in_both_arguments(hips, oracles, key, name):
    """G1 = G2 for types a, a1, e: e(P, P), e(Q, P) and e(P, -P) = e(P, P)^-1 against the oracle."""
    v = golden(name)
    H, O = hips[key], oracles[key]
    n = 4
    P, Q = v.g1[:n], v.g2[:n]
    fb = H.length_in_bytes_G1 // 2
    q = param_value(key, "p" if key == "a1" else "q")
    negP = P.copy()
    for i in range(n):
        y = int.from_bytes(P[i, fb:].tobytes(), "big")
        negP[i, fb:] = _be((q - y) % q, fb)
    for a, b in ((P, P), (Q, P), (P, negP)):
        assert np.array_equal(H.element_pairing(a, b), O.pairing_batch(a, b))
    ePP, ePnP = H.element_pairing(P, P), H.element_pairing(P, negP)
    one = np.zeros(H.length_in_bytes_GT, np.uint8)
    one[fb - 1] = 1
    assert np.array_equal(H.element_mul_GT(ePP, ePnP), np.tile(one, (n, 1)))
    assert not np.array_equal(ePP, np.tile(one, (n, 1)))           # e(P, P) != 1 on these curves


def test_host_batches_range_split_over_a_device_set(hips):
    """pbc_hip_pairing_use_devices: chunks of a host batch go to the devices of the set in turn (here the
    one GPU of the box listed once, twice and three times -- the multi-device code path, more chunks
    than devices, a ragged last chunk); results are identical to the single-device call."""
    import pbc_amd
    from conftest import _param
    v = golden("d_chain256.vec")
    n = 3 * 131072 + 777                                   # 4 chunks
    idx = np.arange(n)
    g1, g2 = v.g1[(idx // v.n) % v.n], v.g2[idx % v.n]
    H = pbc_amd.Pairing(_param("d159"))
    want = H.element_pairing(g1, g2)
    assert np.array_equal(want[(np.arange(v.n) * (v.n + 1))], v.gt)
    for devs in ([0], [0, 0], [0, 0, 0]):
        H.use_devices(devs)
        assert np.array_equal(H.element_pairing(g1, g2), want), devs
    H.use_devices([0, 0])
    k = 4
    m = 40000
    assert np.array_equal(H.element_prod_pairing(g1[:m * k], g2[:m * k], k),
                          hips["d"].element_prod_pairing(g1[:m * k], g2[:m * k], k))
    H.use_devices([])
    assert np.array_equal(H.element_pairing(g1[:1000], g2[:1000]), want[:1000])
    with pytest.raises(pbc_amd.PbcHipError):
        H.use_devices([99])
    H.clear()


@pytest.mark.parametrize("g", GENERIC_A + GENERIC_OTHER)
def test_type_a_parameter_sets_of_other_sizes(hips, oracles, g):
    """pbc_param_init_a_gen output other than (160, 512): 253-, 498-, 765-bit q and r with negative Solinas
    signs run on the bit-by-bit kernels (16- or 33-word arithmetic): reference vectors, products, cross
    pairs and group operations against the oracle, preprocessing, bilinearity on the device."""
    H, O = hips[g], oracles[g]
    v = golden(FILES_OF[g][0])
    assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt)
    w = golden(FILES_OF[g][2])
    assert np.array_equal(H.element_prod_pairing(w.g1, w.g2, w.k), w.gt)
    i, j = np.meshgrid(np.arange(v.n), np.arange(v.n), indexing="ij")
    g1, g2 = v.g1[i.ravel()], v.g2[j.ravel()]
    got = H.element_pairing(g1, g2)
    sel = [1, 7, 20, 35] if g == "a_224_768" else list(range(0, 36, 3))
    assert np.array_equal(got[sel], O.pairing_batch(g1[sel], g2[sel]))
    if not g.startswith("e_"):                             # type e has no preprocessed form
        pp = H.pp_init(v.g1[2])
        assert np.array_equal(pp.apply(v.g2), got.reshape(v.n, v.n, -1)[2])
    r = param_value(g, "n" if g.startswith("a1_") else "r")
    zl = H.length_in_bytes_Zr
    rng = np.random.default_rng(37)
    Z = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r, zl) for _ in range(v.n)])
    aP = H.element_mul_zn(1, v.g1, Z)
    assert np.array_equal(aP[:2], O.g_mul(1, v.g1[:2], Z[:2]))
    assert np.array_equal(H.element_pairing(aP, v.g2), H.element_pow_zn_GT(v.gt, Z))
    h = H.element_from_hash(1, rng.integers(0, 256, (v.n, 20), dtype=np.uint8))
    assert not H.element_mul_zn(1, h, np.tile(_be(r, zl + 1)[-zl:] if False else _be(r % (1 << (8 * zl)), zl), (v.n, 1))).any()


@pytest.mark.parametrize("g", GENERIC_F)
def test_bn_parameter_sets_of_other_sizes(hips, oracles, g):
    """pbc_param_init_f_gen(256) / (200): 254- and 198-bit BN fields on the 8-word arithmetic -- reference
    vectors, products, cross pairs vs the oracle, G1 / G2 scalar multiplication, hash-to-curve, bilinearity in
    both arguments on the device."""
    H, O = hips[g], oracles[g]
    v = golden(FILES_OF[g][0])
    assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt)
    w = golden(FILES_OF[g][2])
    assert np.array_equal(H.element_prod_pairing(w.g1, w.g2, w.k), w.gt)
    g1, g2 = v.g1[[0, 1, 2, 3, 0]], v.g2[[1, 2, 3, 0, 0]]
    assert np.array_equal(H.element_pairing(g1, g2), O.pairing_batch(g1, g2))
    r = param_value(g, "r")
    zl = H.length_in_bytes_Zr
    rng = np.random.default_rng(43)
    Z = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r, zl) for _ in range(v.n)])
    aP = H.element_mul_zn(1, v.g1, Z)
    assert np.array_equal(aP[:2], O.g_mul(1, v.g1[:2], Z[:2]))
    aQ = H.element_mul_zn(2, v.g2, Z)
    want = H.element_pow_zn_GT(v.gt, Z)
    assert np.array_equal(H.element_pairing(aP, v.g2), want)
    assert np.array_equal(H.element_pairing(v.g1, aQ), want)
    assert np.array_equal(want[:2], O.gt_pow(v.gt[:2], Z[:2]))
    h = H.element_from_hash(1, rng.integers(0, 256, (v.n, 32), dtype=np.uint8))
    assert not H.element_mul_zn(1, h, np.tile(_be(r, zl), (v.n, 1))).any()


def test_type_g_chain_and_products(hips):
    H = hips["g149"]
    v = golden("g149_chain64.vec")
    assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt)
    v = golden("g149_prod4x3.vec")
    assert np.array_equal(H.element_prod_pairing(v.g1, v.g2, v.k), v.gt)


@pytest.mark.parametrize("d", OTHER)
def test_other_type_d_and_g_params_cross_pairs_fq_and_group_ops_vs_oracle(hips, oracles, d):
    H, O = hips[d], oracles[d]
    v = golden(FILES_OF[d][0])
    q, r = (param_value(d, "p"), param_value(d, "n")) if d == "a1" else (param_value(d, "q"), param_value(d, "r"))
    fb, zl = (q.bit_length() + 7) // 8, (r.bit_length() + 7) // 8
    # all 144 (P_i, Q_j) combinations, a whole block plus a ragged tail
    i, j = np.meshgrid(np.arange(v.n), np.arange(v.n), indexing="ij")
    g1, g2 = v.g1[i.ravel()], v.g2[j.ravel()]
    assert np.array_equal(H.element_pairing(g1, g2), O.pairing_batch(g1, g2))
    rng = np.random.default_rng(31)
    xs = [int.from_bytes(rng.bytes(fb), "big") % q for _ in range(200)] + [0, 1, q - 1, 2 ** (8 * fb) - 1]
    ys = [int.from_bytes(rng.bytes(fb), "big") % q for _ in range(200)] + [q - 1, 0, q - 1, 2 ** (8 * fb) - 1]
    A, B = np.stack([_be(x, fb) for x in xs]), np.stack([_be(y, fb) for y in ys])
    for op in range(7):
        got, want = H.fq_op(op, A, B), O.fq_op(op, A, B)
        if op == 3:
            keep = np.array([x % q != 0 for x in xs])
            got, want = got[keep], want[keep]
        assert np.array_equal(got, want), "fq op %d" % op
    assert H.length_in_bytes_Zr == zl
    n = v.n
    Z = np.stack([_be(k, zl) for k in [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n - 2)] + [1, r - 1]])
    assert np.array_equal(H.element_mul_zn(1, v.g1, Z), O.g_mul(1, v.g1, Z))
    a, b = v.gt, np.roll(v.gt, 1, axis=0)
    assert np.array_equal(H.element_mul_GT(a, b), O.gt_mul(a, b))
    assert np.array_equal(H.element_pow_zn_GT(a, Z), O.gt_pow(a, Z))
    # bilinearity on the device: e([a]P, Q) == e(P, Q)^a
    assert np.array_equal(H.element_pairing(H.element_mul_zn(1, v.g1, Z), v.g2), H.element_pow_zn_GT(v.gt, Z))


# ---- group operations: element_mul_zn on G1/G2, element_mul / pow_zn on GT ----
R_OF = {"a": R_A, "d": 208617601094290618684641029477488665211553761021,
        "f": 205523667896953300194895899082072403858390252929}


@pytest.mark.parametrize("t,name", [("a", "a_rand32.vec"), ("d", "d_rand32.vec"), ("f", "f_rand16.vec")])
def test_group_ops_vs_oracle(hips, oracles, t, name):
    v = golden(name)
    n = min(v.n, 16)
    rng = np.random.default_rng(23)
    ks = [int.from_bytes(rng.bytes(20), "big") % R_OF[t] for _ in range(n - 2)] + [1, R_OF[t] - 1]
    Z = np.stack([_be(k, 20) for k in ks])
    H, O = hips[t], oracles[t]
    assert H.length_in_bytes_Zr == 20
    assert np.array_equal(H.element_mul_zn(1, v.g1[:n], Z), O.g_mul(1, v.g1[:n], Z))
    if t == "a":
        assert np.array_equal(H.element_mul_zn(2, v.g2[:n], Z), O.g_mul(2, v.g2[:n], Z))
    a, b = v.gt[:n], np.roll(v.gt[:n], 1, axis=0)
    assert np.array_equal(H.element_mul_GT(a, b), O.gt_mul(a, b))
    assert np.array_equal(H.element_pow_zn_GT(a, Z), O.gt_pow(a, Z))


@pytest.mark.parametrize("t,name,n", [("a", "a_chain1024.vec", 1024), ("d", "d_chain256.vec", 256), ("f", "f_chain128.vec", 128)])
def test_bilinearity_entirely_on_gpu(hips, t, name, n):
    """pbc/bilinear.test:24-33 with every operation on the device: e([a]P, Q) == e(P, Q)^a,
    two independent kernel paths (curve scalar multiplication + pairing vs pairing + GT power)."""
    v = golden(name)
    H = hips[t]
    rng = np.random.default_rng(29)
    Z = np.stack([_be(int.from_bytes(rng.bytes(20), "big") % R_OF[t], 20) for _ in range(n)])
    aP = H.element_mul_zn(1, v.g1[:n], Z)
    lhs = H.element_pairing(aP, v.g2[:n])
    rhs = H.element_pow_zn_GT(v.gt[:n], Z)                 # v.gt = e(P_i, Q_i) from the reference
    assert np.array_equal(lhs, rhs)


# ---- element_from_hash and a BLS round trip entirely on the device ---------------------------
@pytest.mark.parametrize("name", ["a_hash32.vec", "a_hash13.vec", "a_hash100.vec"])
def test_from_hash_matches_reference(hip_a, name):
    v = golden(name)
    assert np.array_equal(hip_a.element_from_hash(1, v.g1.reshape(v.n, v.len1)), v.gt)


@pytest.mark.parametrize("key,name", [("d", "d159_hash32.vec"), ("d201", "d201_hash32.vec"),
                                      ("d278027-190-181", "d278027-190-181_hash32.vec"), ("f", "f_hash32.vec"),
                                      ("g149", "g149_hash32.vec"), ("e", "e_hash20.vec"), ("a1", "a1_hash20.vec")])
def test_from_hash_other_types_matches_reference(hips, key, name):
    """curve_from_hash on G1 with Tonelli-Shanks square roots (q = 1 mod 4: d*, e), cofactors (d, g, e, a1)
    and without one (f)."""
    v = golden(name)
    H = hips[key]
    got = H.element_from_hash(1, v.g1.reshape(v.n, v.len1))
    assert np.array_equal(got, v.gt)
    # the hashed points are in G1: [r] P = O, and they pair non-trivially with a G2 element
    r = param_value(key, "n" if key == "a1" else "r")
    zl = H.length_in_bytes_Zr
    assert not H.element_mul_zn(1, got, np.tile(_be(r, zl), (v.n, 1))).any()


@pytest.mark.parametrize("key,name", [("d", "d159_g2mul6.vec"), ("d201", "d201_g2mul6.vec"), ("g149", "g149_g2mul6.vec"),
                                      ("f", "f_g2mul6.vec")])
def test_g2_scalar_multiplication_on_twists(hips, key, name):
    """element_mul_zn on G2 of the asymmetric types vs the reference's outputs, then bilinearity in the
    second argument entirely on the device: e(P, [a]Q) == e(P, Q)^a."""
    v = golden(name)
    H = hips[key]
    aQ = H.element_mul_zn(2, v.g1, v.g2)                   # file: G2 points, scalars, [k]Q
    assert np.array_equal(aQ, v.gt)
    pv = golden({"d": "d_rand32.vec", "d201": "d201_rand12.vec", "g149": "g149_rand16.vec", "f": "f_rand16.vec"}[key])
    P = pv.g1[:v.n]
    assert np.array_equal(H.element_pairing(P, aQ), H.element_pow_zn_GT(H.element_pairing(P, v.g1), v.g2))


@pytest.mark.parametrize("key,name", [("e", "e_g1mul3.vec"), ("d224", "d224_g1mul6.vec")])
def test_g1_scalar_multiplication_wide_fields(hips, key, name):
    v = golden(name)
    assert np.array_equal(hips[key].element_mul_zn(1, v.g1, v.g2), v.gt)


@pytest.mark.parametrize("key,name", [("a", "a_compress12.vec"), ("d", "d159_compress12.vec"),
                                      ("d278027-190-181", "d278027-190-181_compress12.vec"), ("f", "f_compress12.vec"),
                                      ("g149", "g149_compress12.vec"), ("e", "e_compress4.vec")])
def test_compressed_points_match_reference(hips, key, name):
    """element_to_bytes_compressed / element_from_bytes_compressed (ecc/curve.c:762-815) on G1."""
    v = golden(name)
    H = hips[key]
    assert np.array_equal(H.element_to_bytes_compressed(1, v.g1), v.gt)
    assert np.array_equal(H.element_from_bytes_compressed(1, v.gt), v.g1)
    flipped = v.gt.copy()
    flipped[:, -1] ^= 1                                    # the other root: (x, -y)
    neg = H.element_from_bytes_compressed(1, flipped)
    assert np.array_equal(neg[:, :v.len1 // 2], v.g1[:, :v.len1 // 2]) and not np.array_equal(neg, v.g1)
    assert np.array_equal(H.element_to_bytes_compressed(1, neg), flipped)


def test_asymmetric_bls_round_trip_on_gpu(hips):
    """BLS on the BN curve (type f), everything on the device: sk_i, pk_i = [sk_i] g2 (G2 twist),
    h_i = H(m_i) in G1, sig_i = [sk_i] h_i; verify e(sig_i, g2) == e(h_i, pk_i); a forged one fails."""
    H = hips["f"]
    g2 = golden("f_g2mul6.vec").g1[:1]                     # a G2 element from the reference
    n = 64
    rng = np.random.default_rng(41)
    r = param_value("f", "r")
    zl = H.length_in_bytes_Zr
    sk = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % (r - 1) + 1, zl) for _ in range(n)])
    pk = H.element_mul_zn(2, np.tile(g2, (n, 1)), sk)
    h = H.element_from_hash(1, rng.integers(0, 256, (n, 32), dtype=np.uint8))
    sig = H.element_mul_zn(1, h, sk)
    sig[5] = sig[6]                                        # forgery
    lhs = H.element_pairing(sig, np.tile(g2, (n, 1)))
    rhs = H.element_pairing(h, pk)
    ok = (lhs == rhs).all(axis=1)
    assert ok.sum() == n - 1 and not ok[5]


def test_bls_sign_verify_batch_on_gpu(hip_a, oracle_a):
    """example/bls.c:41-117 as a batch: h = from_hash(msg), sig = h^sk, pk = g^sk,
    verify e(sig, g) == e(h, pk).  Every group operation and pairing runs on the GPU; the fixed
    second... first arguments use preprocessed pairings.  One forged signature must fail."""
    v = golden("a_chain1024.vec")
    n = 500
    rng = np.random.default_rng(31)
    g = v.g2[0]                                           # system parameter g in G2 (= G1 for type a)
    sk = int.from_bytes(rng.bytes(20), "big") % R_A
    SK = np.tile(_be(sk, 20), (n, 1))
    pk = hip_a.element_mul_zn(2, g[None, :], SK[:1])[0]
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    h = hip_a.element_from_hash(1, msgs)
    sig = hip_a.element_mul_zn(1, h, SK)
    sig[7] = hip_a.element_mul_zn(1, h[7:8], np.tile(_be(sk + 1, 20), (1, 1)))[0]    # forgery
    # e(sig_i, g) and e(h_i, pk): type a is symmetric, so both are pp_apply with a fixed argument
    lhs = hip_a.pp_init(g).apply(sig)
    rhs = hip_a.pp_init(pk).apply(h)
    ok = (lhs == rhs).all(axis=1)
    assert not ok[7] and ok.sum() == n - 1
    # spot check against the CPU oracle
    assert np.array_equal(lhs[:3], oracle_a.pairing_batch(sig[:3], np.tile(g, (3, 1))))
    assert np.array_equal(h[:2], oracle_a.g_mul(1, h[:2], np.tile(_be(1, 20), (2, 1))))   # valid curve points


@pytest.mark.parametrize("key,name,exact", XONLY)
def test_x_only_points_match_reference(hips, key, name, exact):
    """element_to_bytes_x_only / element_from_bytes_x_only (ecc/curve.c:821-836) on G1"""
    v = golden(name)
    H = hips[key]
    check_x_only(lambda p: H.element_to_bytes_x_only(1, p), lambda x: H.element_from_bytes_x_only(1, x), v, exact,
                 0 if exact else param_value(key, "q"))


@pytest.mark.parametrize("key,name", G2_HASH)
def test_from_hash_on_the_twists_matches_reference(hips, key, name):
    """element_from_hash on G2 of types d, g, f (square roots in F_q^d / F_q^2 on the device)"""
    v = golden(name)
    H = hips[key]
    pts = H.element_from_hash(2, v.g1)
    assert np.array_equal(pts, v.gt)
    # the hashed points lie on the twist: multiplying by 1 keeps them (off-curve input would give O = zeros)
    one = np.zeros((v.n, H.length_in_bytes_Zr), np.uint8)
    one[:, -1] = 1
    assert np.array_equal(H.element_mul_zn(2, pts, one), pts)


@pytest.mark.parametrize("key,name", G2_COMPRESS)
def test_compressed_points_on_the_twists_match_reference(hips, key, name):
    v = golden(name)
    H = hips[key]
    assert np.array_equal(H.element_to_bytes_compressed(2, v.g1), v.gt)
    assert np.array_equal(H.element_from_bytes_compressed(2, v.gt), v.g1)
    flipped = v.gt.copy()
    flipped[:, -1] ^= 1
    neg = H.element_from_bytes_compressed(2, flipped)
    assert np.array_equal(neg[:, :v.len1 // 2], v.g1[:, :v.len1 // 2]) and not np.array_equal(neg, v.g1)
    assert np.array_equal(H.element_to_bytes_compressed(2, neg), flipped)


@pytest.mark.parametrize("key,name,exact", G2_XONLY)
def test_x_only_points_on_the_twists_match_reference(hips, key, name, exact):
    v = golden(name)
    H = hips[key]
    check_x_only_g2(lambda p: H.element_to_bytes_x_only(2, p), lambda x: H.element_from_bytes_x_only(2, x), v, exact,
                    param_value(key, "q"), H.length_in_bytes_Fq)


@pytest.mark.parametrize("pname", ["a", "d159", "f", "g149"])
def test_from_hash_through_the_glue(pname):
    """element_from_hash_batch of integration/pbc_hip_glue.c on G1 and G2 element_t arrays vs the reference's own
    element_from_hash (glue_test ... hash)"""
    import os
    import subprocess
    import pbc_amd
    if not os.path.exists(oracle.GLUE_TEST):
        pytest.skip("oracle/_ref/glue_test not built (needs /root/reference at build time)")
    env = dict(os.environ, PBC_HIP_LIB=pbc_amd.LIB_PATH)
    r = subprocess.run([oracle.GLUE_TEST, os.path.join(pbc_amd.PARAM_DIR, pname + ".param"), "8", "hash"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("key,hlen", [("a", 20), ("a", 70), ("d", 32), ("f", 24), ("g149", 32), ("d224", 28), ("e_160_400", 20)])
def test_from_hash_and_point_formats_on_fresh_digests_vs_oracle(hips, oracles, key, hlen):
    """seeded digests no fixture holds, at a size that exercises the x <- x^2 + 1 retries many times over:
    element_from_hash, compressed and x-only forms of the hashed points vs the oracle"""
    rng = np.random.default_rng(hlen * 131 + len(key))
    n = 96 if key in ("a", "e_160_400") else 400           # the oracle pays two Fermat inversions per cofactor bit
    D = rng.integers(0, 256, (n, hlen), dtype=np.uint8)
    H, O = hips[key], oracles[key]
    pts = H.element_from_hash(1, D)
    assert np.array_equal(pts, O.from_hash(D))
    c = H.element_to_bytes_compressed(1, pts)
    assert np.array_equal(c, O.point_format(0, pts))
    assert np.array_equal(H.element_from_bytes_compressed(1, c), pts)
    fb = H.length_in_bytes_Fq
    back = H.element_from_bytes_x_only(1, H.element_to_bytes_x_only(1, pts))
    assert np.array_equal(back[:, :fb], pts[:, :fb])
    assert np.array_equal(H.element_to_bytes_compressed(1, back)[:, :fb], c[:, :fb])


@pytest.mark.parametrize("key,hlen", [("d", 32), ("d224", 20), ("f", 32), ("f_256", 21), ("g149", 32)])
def test_twist_hashing_and_compression_on_fresh_digests_vs_oracle(hips, oracles, key, hlen):
    """element_from_hash and the compressed form on the G2 twists, 200 seeded digests per curve vs the oracle"""
    rng = np.random.default_rng(hlen * 17 + len(key))
    D = rng.integers(0, 256, (200, hlen), dtype=np.uint8)
    H, O = hips[key], oracles[key]
    pts = H.element_from_hash(2, D)
    assert np.array_equal(pts, O.from_hash_g2(D))
    c = H.element_to_bytes_compressed(2, pts)
    assert np.array_equal(c, O.point_format_g2(0, pts))
    assert np.array_equal(H.element_from_bytes_compressed(2, c), pts)


# ---- inputs outside the order-r subgroup: curve_from_bytes (ecc/curve.c:609-623) accepts every point of the curve ----
WHOLE_CURVE = ["a_full12.vec", "a_prodfull3x4.vec", "d159_full12.vec", "d159_prodfull3x4.vec", "g149_full12.vec",
               "g149_prodfull3x4.vec", "f_full12.vec", "f_prodfull3x4.vec", "a1_full4.vec", "a1_prodfull3x4.vec", "d224_full8.vec"]


@pytest.mark.parametrize("name", WHOLE_CURVE)
def test_whole_curve_points_match_reference(hips, oracles, name):
    """Points generated WITHOUT the cofactor multiplication (ref_tool gen ... fullorder = curve_random_no_cofac_solvefory,
    ecc/curve.c:405-422): they lie outside the order-r subgroup wherever the curve has a cofactor (a, a1, d, g; for f the
    twist points).  The Jacobian Miller loops drop factors of F_q^* along the way; that argument must hold for these
    inputs too.  Singles and products against the reference's outputs, then a cross product of the points against the
    oracle.  (Type e is left out on purpose: for k = 1 the reference's value depends on the auxiliary point R it draws at
    random when the pairing is initialised unless r P = O -- `ref_tool rdep e.param` shows two objects disagreeing.)"""
    from conftest import key_of
    v = golden(name)
    key = key_of(name)
    H, O = hips[key], oracles[key]
    if v.k == 1:
        assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt)
        i, j = np.divmod(np.arange(min(v.n * v.n, 24)), v.n)
        assert np.array_equal(H.element_pairing(v.g1[i], v.g2[j]), O.pairing_batch(v.g1[i], v.g2[j]))
    else:
        assert np.array_equal(H.element_prod_pairing(v.g1, v.g2, v.k), v.gt)
        # the same terms with the subgroup fixture's points mixed in, against the oracle
        assert np.array_equal(H.element_prod_pairing(v.g1[::-1], v.g2, v.k), O.prod_pairing_batch(v.g1[::-1], v.g2, v.k))


@pytest.mark.parametrize("key,name,group", [("a", "a_g1mulfull6.vec", 1), ("a", "a_g2mulfull6.vec", 2), ("d", "d159_g1mulfull6.vec", 1),
                                            ("g149", "g149_g1mulfull6.vec", 1), ("e", "e_g1mulfull6.vec", 1)])
def test_scalar_multiplication_on_whole_curve_points(hips, key, name, group):
    v = golden(name)
    assert np.array_equal(hips[key].element_mul_zn(group, v.g1, v.g2), v.gt)


def test_group_law_is_complete(hip_a, oracle_a):
    """element_mul_zn on points of order 2, 3, 4, 6, 12 (type a: 12 | h) and with scalars >= r: the ladder meets
    R = P (a doubling), R = -P and R = O; every case against the oracle's affine group law."""
    v = golden("a_full12.vec")
    pts = [oracle_a.g_mul(1, v.g1[:4], np.tile(_be((Q_A + 1) // m, 64), (4, 1))) for m in (2, 3, 4, 6, 12)]
    P = np.concatenate(pts)
    assert sum(bool(p.any()) for p in P) >= 8
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 13, R_A - 1, R_A, R_A + 1, R_A + 5, 2 ** 160 - 1):
        Z = np.tile(_be(k, 20), (len(P), 1))
        assert np.array_equal(hip_a.element_mul_zn(1, P, Z), oracle_a.g_mul(1, P, Z)), k
    Z = np.stack([_be(R_A + 1000 * i + 7, 20) for i in range(12)])
    assert np.array_equal(hip_a.element_mul_zn(1, v.g1, Z), oracle_a.g_mul(1, v.g1, Z))


@pytest.mark.parametrize("key,name", [("a", "a_rand32.vec"), ("d", "d_rand32.vec"), ("f", "f_rand16.vec"), ("g149", "g149_rand16.vec"),
                                      ("e", "e_rand6.vec"), ("a1", "a1_rand6.vec")])
def test_all_zero_records_give_the_identity(hips, oracles, key, name):
    """include/pbc_hip.h, "zero-filled records": an all-zero G1 / G2 record gives the GT identity in element_pairing,
    forces a product to the identity, and is a fixed point of element_mul_zn.  (Zeros are what element_to_bytes writes
    for O.  Where b != 0 they are off the curve, i.e. O again; on y^2 = x^3 + x (types a, a1) they are the 2-torsion
    point (0, 0), whose pairing value is 1 as well because 2 is coprime to r -- the reference divides by zero there.)"""
    v = golden(name)
    H, O = hips[key], oracles[key]
    n = 4
    one = H.element_pairing(np.zeros((1, H.length_in_bytes_G1), np.uint8), v.g2[:1])[0]
    assert np.array_equal(one, O.pairing_batch(np.zeros((1, H.length_in_bytes_G1), np.uint8), v.g2[:1])[0])
    assert np.array_equal(H.element_mul_GT(one[None, :], v.gt[:1]), v.gt[:1])           # it IS the identity of GT
    z1, z2 = np.zeros_like(v.g1[:n]), np.zeros_like(v.g2[:n])
    for g1, g2 in ((z1, v.g2[:n]), (v.g1[:n], z2), (z1, z2)):
        assert np.array_equal(H.element_pairing(g1, g2), np.tile(one, (n, 1)))
    g1, g2 = v.g1[:n].copy(), v.g2[:n].copy()
    g1[1] = 0                                              # unit 0 = terms 0, 1 holds a zero record; unit 1 = terms 2, 3 does not
    got = H.element_prod_pairing(g1, g2, 2)
    assert np.array_equal(got[0], one)
    assert np.array_equal(got, O.prod_pairing_batch(g1, g2, 2)) and not np.array_equal(got[1], one)
    zl = H.length_in_bytes_Zr
    Z = np.tile(_be(5, zl), (n, 1))
    assert not H.element_mul_zn(1, z1, Z).any()


def test_bench_two_ranks_share_the_gpu(tmp_path):
    """bench.py --gpus 2 under torch.distributed.run with both ranks on cuda:0 (PBC_BENCH_SAME_DEVICE=1) and gloo for the
    barrier / clock: the multi-rank path of the benchmark (rank-0 build, range split, max-over-ranks clock, per-rank
    kernel times) runs the HIP library before a driver launches it on 8 GPUs."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PBC_BENCH_SAME_DEVICE="1", PBC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 << 16
    assert len(j["per_rank_kernel_ms"]) == 2 and all(x > 0 for x in j["per_rank_kernel_ms"])
    assert j["value"] > 0 and j["scaling"] == "weak"


@pytest.mark.parametrize("how", ["LD_PRELOAD", "link-wrap"])
@pytest.mark.parametrize("pname", ["a", "d159", "f"])
def test_unmodified_program_runs_its_pairings_on_the_gpu(how, pname):
    """The reference's own example/bls.c, compiled WITHOUT a source change (oracle/Makefile `preload`), run as
    `LD_PRELOAD=libpbc_hip_preload.so ./bls_example <param>` against the stock shared libpbc, and the same source linked
    statically with -Wl,--wrap=pairing_init_pbc_param: integration/pbc_hip_preload.c interposes pairing_init_pbc_param
    (ecc/pairing.c:74-86) and attaches the GPU behind pairing->map / pp_*.  The program's own verdicts must hold
    ("signature verifies", "random signature doesn't verify") and the call counters must show that its pairings ran
    through libpbc_hip.so."""
    import os
    import re
    import subprocess
    import pbc_amd
    exe = oracle.BLS_EXAMPLE if how == "LD_PRELOAD" else oracle.BLS_EXAMPLE_WRAPPED
    if not (os.path.exists(exe) and os.path.exists(oracle.PRELOAD_LIB)):
        pytest.skip("oracle/_ref drop-in demo not built (needs /root/reference at build time)")
    env = dict(os.environ, PBC_HIP_LIB=pbc_amd.LIB_PATH, PBC_HIP_VERBOSE="1")
    if how == "LD_PRELOAD":
        env["LD_PRELOAD"] = oracle.PRELOAD_LIB
    r = subprocess.run([exe, os.path.join(pbc_amd.PARAM_DIR, pname + ".param")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "signature verifies" in r.stdout and "random signature doesn't verify" in r.stdout
    m = re.search(r"on the GPU: (\d+) element_pairing, (\d+) element_prod_pairing, (\d+) pairing_pp_init, (\d+) pairing_pp_apply", r.stderr)
    assert m, r.stderr[-1500:]
    assert int(m.group(1)) >= 4                                    # bls.c calls element_pairing at lines 70, 75, 100, 117, ...
    # and without the interposer the very same binary stays on the CPU
    if how == "LD_PRELOAD":
        env.pop("LD_PRELOAD")
        r2 = subprocess.run([exe, os.path.join(pbc_amd.PARAM_DIR, pname + ".param")], capture_output=True, text=True, env=env, timeout=600)
        assert r2.returncode == 0 and "on the GPU" not in r2.stderr


def test_type_a_with_a_1024_bit_field(hips, oracles):
    """pbc_param_init_a_gen(160, 1024): 1024-bit q on the 33-word arithmetic, 864-bit cofactor (init refused
    cofactors above 768 bits).  Pairings, products with identity inputs, element_from_hash and [r] H = O."""
    H, O = hips["a_160_1024"], oracles["a_160_1024"]
    v, pr, h = golden("a_160_1024_rand4.vec"), golden("a_160_1024_prod3x3_edge.vec"), golden("a_160_1024_hash20.vec")
    assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt)
    assert np.array_equal(H.element_prod_pairing(pr.g1, pr.g2, pr.k), pr.gt)
    pts = H.element_from_hash(1, h.g1)
    assert np.array_equal(pts, h.gt)
    r = param_value("a_160_1024", "r")
    zl = H.length_in_bytes_Zr
    assert not H.element_mul_zn(1, pts, np.tile(_be(r, zl), (h.n, 1))).any()
    i, j = np.divmod(np.arange(8), 4)
    assert np.array_equal(H.element_pairing(v.g1[i], v.g2[j]), O.pairing_batch(v.g1[i], v.g2[j]))


def test_host_buffers_pinned_in_place_and_staged(hips):
    """Host-buffer entry points: page-locked caller buffers are read / written by the kernels in place (zero-copy),
    pageable ones and "hip_zero_copy 0" go through staged chunk buffers -- same bytes, singles and products, ragged n."""
    import ctypes
    import torch
    import pbc_amd
    from conftest import _param
    L = pbc_amd.lib()
    for key, name, pname in (("a", "a_chain1024.vec", "a"), ("d", "d_chain256.vec", "d159"), ("f", "f_chain128.vec", "f")):
        v = golden(name)
        staged = pbc_amd.Pairing(_param(pname) + "hip_zero_copy 0\nhip_host_chunk 100\n")
        for k, n in ((1, 333), (3, 41)):
            i = np.arange(n * k) % v.n
            j = (np.arange(n * k) * 7) % v.n
            g1, g2 = np.ascontiguousarray(v.g1[i]), np.ascontiguousarray(v.g2[j])
            want = hips[key].element_prod_pairing(g1, g2, k)                     # pageable numpy buffers: staged
            h1, h2 = torch.from_numpy(g1).pin_memory(), torch.from_numpy(g2).pin_memory()
            out = torch.zeros(n, v.lenT, dtype=torch.uint8).pin_memory()
            for P in (hips[key], staged):                                         # pinned: in place / staged in chunks of 100
                out.zero_()
                assert L.pbc_hip_element_prod_pairing_batch(P._h, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(h1.data_ptr()),
                                                            ctypes.c_void_p(h2.data_ptr()), n, k) == 0
                assert np.array_equal(out.numpy(), want)
            if k == 1:                                                            # pairs (t, t) are the fixture's own
                d = np.nonzero(i == j)[0]
                assert len(d) and np.array_equal(want[d], v.gt[i[d]])
        staged.clear()
        if key in ("a", "d"):                                                     # pairing_pp_apply: the same two routes
            pp = hips[key].pp_init(v.g1[1])
            g2 = np.ascontiguousarray(v.g2[np.arange(77) % v.n])
            want = pp.apply(g2)
            h2 = torch.from_numpy(g2).pin_memory()
            out = torch.zeros(77, v.lenT, dtype=torch.uint8).pin_memory()
            assert L.pbc_hip_pairing_pp_apply_batch(pp._h, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(h2.data_ptr()), 77) == 0
            assert np.array_equal(out.numpy(), want) and np.array_equal(want[1], v.gt[1])
            pp.clear()


def test_two_objects_with_the_same_word_count_run_concurrently(hips):
    """d159.param and f.param are both 5-word fields: in round 1 their constants shared process-global __constant__
    symbols, so two objects on two streams raced.  The constants now travel in each launch's own argument block: a d159
    and an f batch (and a second d159 object with a different stream) enqueued back to back on separate streams, several
    rounds, every result bit-exact."""
    import torch
    import pbc_amd
    from conftest import _param
    vd, vf = golden("d_chain256.vec"), golden("f_chain128.vec")
    Hd, Hf = hips["d"], hips["f"]
    Hd2 = pbc_amd.Pairing(_param("d201"))                  # 7 words: a third constant block in flight
    v2 = golden("d201_rand12.vec")
    n = 1 << 14
    def dev(a, m):
        return torch.from_numpy(np.tile(a, (-(-m // len(a)), 1))[:m].copy()).cuda()
    d1, d2, f1, f2, e1, e2 = dev(vd.g1, n), dev(vd.g2, n), dev(vf.g1, n), dev(vf.g2, n), dev(v2.g1, n), dev(v2.g2, n)
    od = torch.empty(n, vd.lenT, dtype=torch.uint8, device="cuda")
    of = torch.empty(n, vf.lenT, dtype=torch.uint8, device="cuda")
    oe = torch.empty(n, v2.lenT, dtype=torch.uint8, device="cuda")
    sd, sf, se = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(4):
        Hd.element_pairing_dev(od.data_ptr(), d1.data_ptr(), d2.data_ptr(), n, sd.cuda_stream)
        Hf.element_pairing_dev(of.data_ptr(), f1.data_ptr(), f2.data_ptr(), n, sf.cuda_stream)
        Hd2.element_pairing_dev(oe.data_ptr(), e1.data_ptr(), e2.data_ptr(), n, se.cuda_stream)
    torch.cuda.synchronize()
    assert (od.cpu().numpy().reshape(n // vd.n, vd.n, -1) == vd.gt[None]).all()
    assert (of.cpu().numpy().reshape(n // vf.n, vf.n, -1) == vf.gt[None]).all()
    m = n // v2.n * v2.n
    assert (oe[:m].cpu().numpy().reshape(m // v2.n, v2.n, -1) == v2.gt[None]).all()
    Hd2.clear()


def test_product_workspaces_stay_bounded_over_many_streams(hips):
    """The type a / d product kernels keep one workspace per (device, stream); a caller that enqueues on ever new
    streams must not accumulate them: twelve short-lived streams against the table's eight entries -- the
    least recently used buffer is evicted after a device synchronisation -- every product bit-exact, then
    pbc_hip_pairing_release_workspaces and one more launch."""
    import torch
    for key, name in (("a", "a_prod16x4.vec"), ("d", "d_prod16x4.vec")):
        H, w = hips[key], golden(name)
        reps = 64
        g1 = torch.from_numpy(np.tile(w.g1.reshape(-1, w.len1), (reps, 1))).cuda()
        g2 = torch.from_numpy(np.tile(w.g2.reshape(-1, w.len2), (reps, 1))).cuda()
        n = reps * len(w.gt)
        outs = [(torch.zeros(n, w.lenT, dtype=torch.uint8, device="cuda"), torch.cuda.Stream()) for i in range(12)]
        torch.cuda.synchronize()         # the fills run on torch's current stream, the launches below on their own
        for o, st in outs:
            H.element_prod_pairing_dev(o.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, w.k, st.cuda_stream)
        torch.cuda.synchronize()
        for o, _ in outs:
            assert (o.cpu().numpy().reshape(reps, len(w.gt), -1) == w.gt[None]).all()
        H.release_workspaces()
        o = torch.zeros(n, w.lenT, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        H.element_prod_pairing_dev(o.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, w.k, 0)
        torch.cuda.synchronize()
        assert (o.cpu().numpy().reshape(reps, len(w.gt), -1) == w.gt[None]).all()


@pytest.mark.parametrize("pname", ["a", "d159", "f"])
def test_pairing_pp_lifetimes_through_the_glue(pname):
    """Orders stock PBC tolerates: a pairing_pp_t made on the GPU and cleared AFTER pbc_hip_detach or after
    pairing_clear; a pairing_pp_t made on the CPU while the hooks are still installed for an older one; re-attaching;
    element_prod_pairing_batch with k = 0 (empty products = 1, no division by zero).  glue_test ... lifetime."""
    import os
    import subprocess
    import pbc_amd
    if not os.path.exists(oracle.GLUE_TEST):
        pytest.skip("oracle/_ref/glue_test not built (needs /root/reference at build time)")
    env = dict(os.environ, PBC_HIP_LIB=pbc_amd.LIB_PATH)
    r = subprocess.run([oracle.GLUE_TEST, os.path.join(pbc_amd.PARAM_DIR, pname + ".param"), "8", "lifetime"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("key,name", [("a", "a_finalpow6.vec"), ("d", "d159_finalpow6.vec"), ("f", "f_finalpow6.vec"),
                                      ("g149", "g149_finalpow6.vec"), ("e", "e_finalpow3.vec"), ("a1", "a1_finalpow3.vec")])
def test_finalpow_matches_reference(hips, oracles, key, name):
    """pbc_hip_finalpow_batch = pairing->finalpow (include/pbc_pairing.h:41): reference vectors, then fresh elements of
    GT's underlying field (products of the fixture's inputs) against the oracle, and idempotence up to the known power:
    finalpow maps into the order-r subgroup, so a second application raises to (q^k - 1)/r again = the same as
    element_pow by that exponent -- checked through finalpow(x y) = finalpow(x) finalpow(y)."""
    v = golden(name)
    H, O = hips[key], oracles[key]
    assert np.array_equal(H.finalpow(v.g1), v.gt)
    x, y = v.g1, np.roll(v.g1, 1, axis=0)
    xy = O.gt_mul(x, y)
    got = H.finalpow(xy)
    assert np.array_equal(got, O.finalpow(xy))
    assert np.array_equal(got, H.element_mul_GT(H.finalpow(x), H.finalpow(y)))      # a homomorphism
    assert np.array_equal(H.finalpow(_easy_part_is_one(v.g1, key)), O.finalpow(_easy_part_is_one(v.g1, key)))


def _easy_part_is_one(recs, key):
    """GT-format records whose first stage of the final exponentiation gives +-1 (purely "real" elements; for the polymod
    towers also equal coefficients, which is what element_from_hash produces on GT: a product of proper subfields).  The
    single-inversion final exponentiations divide by the "imaginary" part there unless they special-case it."""
    x = recs.copy()
    lt = x.shape[1]
    if key in ("a", "a1"):
        x[:, lt // 2:] = 0
    elif key in ("d", "g149"):
        d = 3 if key == "d" else 5
        fb = lt // (2 * d)
        for i in range(1, d):
            x[:, i * fb:(i + 1) * fb] = x[:, :fb]
            x[:, (d + i) * fb:(d + i + 1) * fb] = x[:, d * fb:(d + 1) * fb]
    elif key == "f":
        x[:, lt // 6:] = 0                       # an element of F_q^2
    return x


@pytest.mark.gpu
def test_a1_preprocessed_pairings_on_a_batch_that_fills_two_waves_per_simd(hips):
    """a1.param pairing_pp_apply at 2^17 units (round 5: 256-lane workgroups, two waves per SIMD, f^2 of a step in LDS)
    against element_pairing of the same device buffers, and against the reference fixture where the arguments are its own."""
    import torch
    v = golden("a1_chain8.vec")
    H = hips["a1"]
    n, D, LT = 1 << 17, v.n, H.length_in_bytes_GT
    st = torch.cuda.current_stream().cuda_stream
    g2 = torch.from_numpy(v.g2).cuda()[torch.arange(n, device="cuda") % D].contiguous()
    g1 = torch.from_numpy(np.tile(v.g1[3], (n, 1))).cuda()
    pp = H.pp_init(v.g1[3])
    got = torch.empty(n, LT, dtype=torch.uint8, device="cuda")
    want = torch.empty(n, LT, dtype=torch.uint8, device="cuda")
    pp.apply_dev(got.data_ptr(), g2.data_ptr(), n, st)
    H.element_pairing_dev(want.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, st)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert np.array_equal(got[3].cpu().numpy(), v.gt[3])
    pp.clear()


# ---- the reference's montfp limb images as the exchange format (round 6) ---------------------------------------------
@pytest.mark.parametrize("name", ["a_rand32.vec", "a_edge20.vec", "d_rand32.vec", "d_edge20.vec", "f_rand16.vec", "f_edge10.vec", "a_prod16x4.vec",
                                  "g149_rand16.vec", "d201_rand12.vec", "a1_rand6.vec", "e_rand6.vec"])
def test_limb_image_entry_points_match_the_reference_vectors(hips, name):
    """pbc_hip_element_prod_pairing_batch_limbs: inputs and outputs as t little-endian 64-bit limbs of x 2^(64 t) mod q per F_q
    coordinate (what a montfp element holds, arith/montfp.c:36-39) -- the reference's vectors, off-curve and zero records
    included, re-encoded on the host; the results decoded and compared with the reference's bytes"""
    v = golden(name)
    P = hips[key_of(name)]
    t = -(-P.field_order.bit_length() // 64)
    assert P.limb_image_bytes == 8 * t
    got = P.element_prod_pairing_limbs(P.to_limb_images(v.g1), P.to_limb_images(v.g2), v.k)
    assert np.array_equal(P.from_limb_images(got), v.gt)
    # the images are canonical: every coordinate below q
    q, w = P.field_order, 8 * t
    assert all(int.from_bytes(got[i, c * w:(c + 1) * w].tobytes(), "little") < q for i in range(min(4, len(got))) for c in range(got.shape[1] // w))


def test_glue_batch_call_of_several_chunks():
    """integration/pbc_hip_glue.c: element_pairing_batch over three chunks of 131072 pairs with sixteen conversion threads;
    glue_test compares 17 units spread over the chunks with the reference's CPU pairing"""
    import os
    import subprocess
    import pbc_amd
    if not os.path.exists(oracle.GLUE_TEST):
        pytest.skip("oracle/_ref/glue_test not built (needs /root/reference at build time)")
    env = dict(os.environ, PBC_HIP_LIB=pbc_amd.LIB_PATH)
    r = subprocess.run([oracle.GLUE_TEST, os.path.join(pbc_amd.PARAM_DIR, "a.param"), "300000", "bench"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "equals the CPU pairing" in r.stdout, r.stdout + r.stderr


def test_limb_image_calls_in_a_row_from_worker_threads(hips):
    """the shape of the glue's batch pipeline: eight host-buffer calls of 131072 units each on page-locked buffers, every call
    from a fresh thread, results compared in full with one wire-format call"""
    import ctypes
    import threading
    import torch
    import pbc_amd
    P = hips["a"]
    v = golden("a_chain1024.vec")
    n, CH = 8 * 131072, 131072
    m = 1024
    I1, I2 = P.to_limb_images(v.g1[:m]), P.to_limb_images(v.g2[(np.arange(m) * 7 + 3) % v.n])
    want1 = P.from_limb_images(P.element_prod_pairing_limbs(I1, I2, 1))
    assert np.array_equal(want1, P.element_pairing(v.g1[:m], v.g2[(np.arange(m) * 7 + 3) % v.n]))
    h1 = torch.from_numpy(np.tile(I1, (n // m, 1))).pin_memory()
    h2 = torch.from_numpy(np.tile(I2, (n // m, 1))).pin_memory()
    out = torch.zeros(n, I1.shape[1], dtype=torch.uint8).pin_memory()
    L = pbc_amd.lib()
    rcs = []

    def call(lo):
        rcs.append(L.pbc_hip_element_prod_pairing_batch_limbs(P._h, ctypes.c_void_p(out.data_ptr() + lo * out.shape[1]),
                                                              ctypes.c_void_p(h1.data_ptr() + lo * h1.shape[1]),
                                                              ctypes.c_void_p(h2.data_ptr() + lo * h2.shape[1]), CH, 1))
    for lo in range(0, n, CH):
        t = threading.Thread(target=call, args=(lo,))
        t.start()
        t.join()
    assert rcs == [0] * 8
    got = out.numpy().reshape(n // m, m, -1)
    first = got[0]
    assert np.array_equal(P.from_limb_images(first), want1)
    bad = [i for i in range(n // m) if not np.array_equal(got[i], first)]
    assert not bad, "blocks of 1024 units that differ from the first: %s" % bad[:8]

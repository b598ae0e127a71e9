"""
Full-size GPU parity (BASELINE.json sizes) through size-independent properties, plus strided oracle samples:
linearity of scalar multiplication, agreement of independent algorithms (windowed ladder vs LDS fixed-base table
vs Pippenger), encode -> decode round trips, and a strided bit-exact comparison with the CPU oracle.
Inputs are generated on the device (torch), so nothing large crosses PCIe.
"""
import numpy as np
import pytest

from oracle import c_oracle as O
from oracle import jubjub_ref as J
from util import pt64

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def env():
    from jubjub_amd import Engine

    eng = Engine(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0x4A55424A5542)
    base = torch.from_numpy(pt64(J.GENERATOR).copy()).to(dev)
    table = eng.fixedbase_table(base)
    yield eng, dev, g, base, table
    table.close()
    eng.close()


def rand_scalars(dev, g, n, bits252=True):
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    if bits252:
        s[:, 31] &= 0x0F
    return s


def sample_vs_oracle(fn_oracle, got, idx, *inputs):
    want = fn_oracle(*[x[idx].cpu().numpy() for x in inputs])
    assert (got[idx].cpu().numpy() == want).all()


def test_varbase_2p20(env):
    eng, dev, g, base, table = env
    n = 1 << 20
    P = eng.fixedbase_mul(table, rand_scalars(dev, g, n))          # random points of the full group
    k1, k2 = rand_scalars(dev, g, n), rand_scalars(dev, g, n)
    k1[:, 31] &= 0x07
    k2[:, 31] &= 0x07                                              # k1 + k2 < 2^252: integer addition, no wrap
    a, b = eng.varbase_mul(k1, P), eng.varbase_mul(k2, P)
    s = np.zeros((n, 32), np.uint8)                                # add the 256-bit integers on the host (cheap)
    carry = np.zeros(n, np.uint16)
    k1h, k2h = k1.cpu().numpy(), k2.cpu().numpy()
    for j in range(32):
        t = k1h[:, j].astype(np.uint16) + k2h[:, j] + carry
        s[:, j] = t & 0xFF
        carry = t >> 8
    c = eng.varbase_mul(torch.from_numpy(s).to(dev), P)
    assert bool((eng.point_add(a, b) == c).all())                  # (k1 + k2) P == k1 P + k2 P for all 2^20 units
    idx = torch.arange(0, n, 1 << 10, device=dev)                  # 2^10 strided sample, bit-exact vs the oracle ladder
    sample_vs_oracle(O.varbase_mul, a, idx, k1, P)
    enc = eng.compress(a)
    back, ok = eng.decompress(enc, 1)
    assert bool(ok.all()) and bool((back == a).all())              # encode -> decode round trip over all outputs
    assert bool(eng.predicate("is_on_curve", a).all())


def test_fixedbase_2p24(env):
    eng, dev, g, base, table = env
    n = 1 << 24
    k1, k2 = rand_scalars(dev, g, n), rand_scalars(dev, g, n)
    a = eng.fixedbase_mul(table, k1)
    idx = torch.arange(0, n, 1 << 12, device=dev)                  # 2^12 strided sample vs the reference ladder (oracle)
    want = O.fixedbase_mul(k1[idx].cpu().numpy(), base.cpu().numpy())
    assert (a[idx].cpu().numpy() == want).all()
    # independent algorithm: windowed var-base ladder on the same base, 2^18 sample
    m = 1 << 18
    pb = base.reshape(1, 64).expand(m, 64).contiguous()
    assert bool((eng.varbase_mul(k1[:m], pb) == a[:m]).all())
    # linearity over the whole batch with Fr arithmetic on the device: (k1 + k2 mod r) G == k1 G + k2 G is only
    # valid on the prime-order subgroup, so use 8G-scaled results: 8 (k1 G) + 8 (k2 G) == 8 ((k1+k2 mod r) G)
    b = eng.fixedbase_mul(table, k2)
    ks = eng.field_binary("fr", "add", k1, k2)
    c = eng.fixedbase_mul(table, ks)
    lhs = eng.mul_by_cofactor(eng.point_add(a, b))
    assert bool((lhs == eng.mul_by_cofactor(c)).all())


def test_msm_2p20(env):
    eng, dev, g, base, table = env
    n = 1 << 20
    S = rand_scalars(dev, g, n)
    P = eng.fixedbase_mul(table, rand_scalars(dev, g, n))
    total = eng.msm(S, P)
    assert bool((total == eng.point_sum(eng.varbase_mul(S, P))).all())       # Pippenger == sum of ladders (independent algorithms)
    h = n // 3
    parts = torch.stack([eng.msm(S[:h], P[:h]), eng.msm(S[h:], P[h:])])
    assert bool((eng.point_sum(parts) == total).all())                         # additivity over a split (the multi-GPU combine)
    m = 1 << 12
    assert (eng.msm(S[:m], P[:m]).cpu().numpy() == O.msm(S[:m].cpu().numpy(), P[:m].cpu().numpy())).all()


def test_decompress_2p22(env):
    eng, dev, g, base, table = env
    n = 1 << 22
    P = eng.fixedbase_mul(table, rand_scalars(dev, g, n))
    enc = eng.compress(P)
    junk = torch.randint(0, 256, (n // 16, 32), dtype=torch.uint8, device=dev, generator=g)
    enc[::16] = junk                                               # 1/16 raw bytes: off-curve, >= q, sign-bit noise
    out, ok = eng.decompress(enc, 1)
    valid = torch.ones(n, dtype=torch.bool, device=dev)
    valid[::16] = False
    assert bool(ok[valid].all()) and bool((out[valid] == P[valid]).all())
    re = eng.compress(out)
    good = ok.bool()
    assert bool((re[good] == enc[good]).all())                     # every accepted encoding re-encodes to itself (canonical)
    assert bool((out[~good] == 0).all())
    idx = torch.arange(0, n, 16, device=dev)[:4096]                # the junk lanes, checked against the oracle decode
    eo, ek = O.decompress(enc[idx].cpu().numpy(), 1)
    assert (ok[idx].cpu().numpy() == ek).all() and (out[idx].cpu().numpy() == eo).all()
    # subgroup variant on a 2^16 slice: decode + [r]P == O, vs oracle on a sample
    m = 1 << 16
    o2, k2 = eng.decompress(enc[:m], 1 | 2)
    eo, ek = O.decompress(enc[:512].cpu().numpy(), 1 | 2)
    assert (k2[:512].cpu().numpy() == ek).all() and (o2[:512].cpu().numpy() == eo).all()


def test_subgroup_check_2p20(env, monkeypatch):
    """2^20 points with a known torsion component: P_i = 8*A_i + (i mod 8)*T for a generator T of the 8-torsion.
    The pairing test must accept exactly the lanes with i = 0 mod 8, and must agree with the ladder mode."""
    from jubjub_amd import Engine

    eng, dev, g, base, table = env
    n = 1 << 20
    A = eng.mul_by_cofactor(eng.fixedbase_mul(table, rand_scalars(dev, g, n)))
    G8 = J.scalar_mul_fast(J.GENERATOR, J.R_MOD)                    # order-8 component of the generator
    assert J.scalar_mul_fast(G8, 4) != J.AFFINE_IDENTITY
    tors = np.stack([pt64(J.scalar_mul_fast(G8, j) if j else J.AFFINE_IDENTITY) for j in range(8)])
    T = torch.from_numpy(tors).to(dev)[torch.arange(n, device=dev) % 8]
    P = eng.point_add(A, T)
    tf = eng.predicate("is_torsion_free", P)
    want = (torch.arange(n, device=dev) % 8 == 0)
    assert bool((tf.bool() == want).all())
    opts = {}
    opts['torsion_check_ladder'] = 1
    e2 = Engine(0, options=opts)
    m = 1 << 17
    assert bool((e2.predicate("is_torsion_free", P[:m]) == tf[:m]).all())
    e2.close()
    enc = eng.compress(P)
    out, ok = eng.decompress(enc, 1 | 2)
    assert bool((ok.bool() == want).all()) and bool((out[want] == P[want]).all()) and bool((out[~want] == 0).all())


def test_msm_2p25_two_passes_and_decompress_2p26(env):
    """Beyond one Pippenger pass (2^24 terms) and at BASELINE config 5's whole size on a single GPU."""
    eng, dev, g, base, table = env
    n = 1 << 25
    S = rand_scalars(dev, g, n)
    P = eng.fixedbase_mul(table, rand_scalars(dev, g, n))
    full = eng.msm(S, P)
    q = 3 * (n // 8)                                               # uneven split: passes of different sizes
    parts = torch.stack([eng.msm(S[:q], P[:q]), eng.msm(S[q:], P[q:])])
    assert bool((eng.point_sum(parts) == full).all())
    del S
    enc = eng.compress(P)
    enc = torch.cat([enc, enc])                                    # 2^26 encodings
    out, ok = eng.decompress(enc, 1 | 4 | 8)
    assert bool(ok.all()) and bool((out[:n] == out[n:]).all())
    idx = torch.arange(0, n, 8191, device=dev)
    assert bool((out[idx] == eng.mul_by_cofactor(P[idx])).all())
    eo, ek = O.decompress(enc[idx[:256]].cpu().numpy(), 1 | 4 | 8)
    assert (out[idx[:256]].cpu().numpy() == eo).all() and (ok[idx[:256]].cpu().numpy() == ek).all()


@pytest.mark.parametrize("fname,which,p", [("fq", O.FQ, J.Q), ("fr", O.FR, J.R_MOD)])
def test_field_ops_2p20_vs_oracle(env, fname, which, p):
    """2^20 random pairs plus structured operands (2^k, 2^k - 1, p - 2^k, saturated 29-bit limb patterns) through
    every field kernel, compared bit-exactly with the C oracle."""
    eng, dev, g, base, table = env
    n = 1 << 20
    a = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g).cpu().numpy()
    b = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g).cpu().numpy()
    special = []
    for k in range(0, 256, 7):
        special += [1 << k, (1 << k) - 1, (p - (1 << k)) % (1 << 256), ((1 << 256) - 1) >> (255 - k)]
    limb_all_ones = sum(0x1FFFFFFF << (29 * i) for i in range(8)) | (0xFFFFFF << 232)
    special += [limb_all_ones, limb_all_ones ^ ((1 << 116) - 1), p - 1, p, p + 1, 2 * p - 1, (1 << 256) - 1, 0, 1]
    sp = np.stack([np.frombuffer(int(x % (1 << 256)).to_bytes(32, "little"), np.uint8) for x in special])
    m = len(sp)
    a[:m], b[:m] = sp, sp[::-1]
    a[m:2 * m], b[m:2 * m] = sp, sp
    for op in ("add", "sub", "mul"):
        assert (eng.field_binary(fname, op, a, b) == O.field_op(which, op, a, b)[0]).all(), op
    for op in ("neg", "square", "double"):
        assert (eng.field_unary(fname, op, a) == O.field_op(which, op, a)[0]).all(), op
    k = 1 << 14                                                   # inversion / sqrt / decode on a 2^14 slice (CPU oracle cost)
    for op in ("invert", "sqrt"):
        out, ok = eng.field_unary_ok(fname, op, a[:k])
        eo, ek = O.field_op(which, op, a[:k])
        assert (ok == ek).all() and (out == eo).all(), op
    out, ok = eng.field_unary_ok(fname, "from_bytes", a)
    eo, ek = O.from_bytes(which, a)
    assert (ok == ek).all() and (out == eo).all()


def test_independent_algorithms_agree_2p22(env):
    """round-3 kernels against their independent siblings over whole batches, compared on the device: the signed comb, the signed
    6-bit window table and the 12-bit gathered table give the same 2^22 points; the constant-time ladder (table in registers,
    2-bit windows) gives the same 2^18 points as the default ladder (per-lane tables, 5-bit windows) on full-group points; the
    composite table (3 x 64-bit scalars, one pass) equals three comb passes"""
    eng, dev, g, base, table = env
    n = 1 << 22
    k = rand_scalars(dev, g, n)
    a = eng.fixedbase_mul(table, k)                                # default = comb
    for wbits in (6, 12):
        t2 = eng.fixedbase_table(base, wbits)
        assert bool(torch.equal(eng.fixedbase_mul(t2, k), a)), wbits
        t2.close()
    m = 1 << 18
    P = a[:m]                                                      # full-group points
    k2 = rand_scalars(dev, g, m, bits252=False)                    # top bits set: ignored by both ladders
    assert bool(torch.equal(eng.varbase_mul_vartime(k2, P), eng.varbase_mul(k2, P)))          # table ladder (5-bit windows) == constant-time ladder (3-bit windows)
    bases = a[:3].contiguous()
    ct = eng.fixedbase_composite_table(bases, [64, 64, 64])
    S = rand_scalars(dev, g, 3 * m).reshape(3, m, 32)
    Sm = S.clone()
    Sm[:, :, 8:] = 0                                               # the low 64 bits: what the composite table uses
    tabs = [eng.fixedbase_table(bases[i], 0) for i in range(3)]
    assert bool(torch.equal(eng.fixedbase_composite_mul(ct, S), eng.fixedbase_multi_mul(tabs, Sm)))
    for t in tabs + [ct]:
        t.close()

"""Parity-check matrices as row-major edge lists (the decoder's only input format).

* `wimax_edges` -- the matrices the reference's `Coder(K, N, rate)` builds from its
  six IEEE 802.16e seed tables (MyLdpc.cpp:52-109).  Independent numpy restatement
  (the oracle and the C++ Coder have their own; tests cross-check all three).
* `dvbs2_profile_edges` -- DVB-S2-profile IRA codes.  The ETSI EN 302 307 Annex B/C
  address tables are in neither the reference nor this image (no network), so the
  table is GENERATED from a seed with the standard's exact structure: 360-column
  groups, check stride q = M/360, the standard's column-degree profile, check-regular
  (every check gets the same number of information edges), staircase parity part.
  Edge count, degrees and memory access pattern equal the standard's code (e.g.
  E = 226 799 at rate 1/2); error-rate results must be labelled "DVB-S2-profile
  surrogate".
* `nr_bg1_profile_edges` -- 5G NR base-graph-1-profile QC code (46 x 68 base, 316
  blocks, BG1's row-weight profile and parity structure), shifts generated from a
  seed for the same reason.
"""
import numpy as np

from .wimax_seeds import NB, SEEDS, Z0

RATE_NAMES = ("1/2", "2/3A", "2/3B", "3/4A", "3/4B", "5/6")
RATE_1_2, RATE_2_3_A, RATE_2_3_B, RATE_3_4_A, RATE_3_4_B, RATE_5_6 = range(6)


def row_major(rows, cols):
    """Sort an edge list into row-major order (edge id = rank)."""
    rows = np.asarray(rows, np.int64)
    cols = np.asarray(cols, np.int64)
    order = np.lexsort((cols, rows))
    r, c = rows[order], cols[order]
    if r.size > 1 and np.any((r[1:] == r[:-1]) & (c[1:] == c[:-1])):
        raise ValueError("duplicate edge")
    return r.astype(np.int32), c.astype(np.int32)


def wimax_dims(rate, N):
    """(K, M, z) of Coder(K, N, rate): z = N/24 (MyLdpc.cpp:55)."""
    z = N // NB
    M = len(SEEDS[rate]) * z
    return N - M, M, z


def wimax_edges(rate, N):
    """Edges of the reference's H for (rate, N), row-major.

    MyLdpc.cpp:86-105: block (seedRow, seedCol) with shift p >= 0 holds ones at
    (r, c) with (z + c - r) % z == p', p' = floor(p*z/96) or p % z for rate 2/3A."""
    seed = np.asarray(SEEDS[rate], np.int64)
    z = N // NB
    if z * NB != N or z <= 0:
        raise ValueError("N must be a positive multiple of 24")
    mb = seed.shape[0]
    pr = np.arange(z)
    rows, cols = [], []
    for sr in range(mb):
        for sc in range(NB):
            p = int(seed[sr, sc])
            if p < 0:
                continue
            p = (p * z) // Z0 if rate != RATE_2_3_A else p % z
            rows.append(sr * z + pr)
            cols.append(sc * z + (pr + p) % z)
    return row_major(np.concatenate(rows), np.concatenate(cols))


# --------------------------------------------------------------------------- DVB-S2

#: (N, K) -> [(column degree, number of 360-column groups), ...] of the information part
DVBS2_PROFILES = {
    (64800, 32400): [(8, 36), (3, 54)],     # rate 1/2:  E = 226 799, check degree 7 (row 0: 6)
    (64800, 58320): [(4, 18), (3, 144)],    # rate 9/10: E = 194 399, check degree 30
    (12960, 6480): [(6, 6), (3, 12)],       # small check-regular instance for tests (not a standard size)
}


def dvbs2_profile_table(N, K, seed=20260101, profile=None):
    """Generate the address table (list of per-group address lists) of a
    DVB-S2-profile IRA code.  Address x of group g means: column 360*g + s has a
    one in row (x + s*q) mod M, s = 0..359 (EN 302 307 sec. 5.3.2 structure)."""
    M = N - K
    if K % 360 or M % 360:
        raise ValueError("K and M must be multiples of 360")
    q = M // 360
    profile = profile or DVBS2_PROFILES[(N, K)]
    groups = [d for d, n in profile for _ in range(n)]
    if len(groups) * 360 != K:
        raise ValueError("degree profile does not cover K columns")
    total = sum(groups)
    if total % q:
        raise ValueError("profile is not check-regular for q=%d" % q)
    per_res = total // q
    rng = np.random.default_rng(seed)
    # every residue class (row mod q) receives exactly per_res addresses
    residues = np.repeat(np.arange(q), per_res)
    table = None
    for _attempt in range(200):
        rng.shuffle(residues)
        table, pos, ok = [], 0, True
        # shifts[(residue)] -> list of (group, a) already placed, for the 4-cycle test
        placed = {}
        for g, d in enumerate(groups):
            res = residues[pos:pos + d]
            pos += d
            addrs = []
            for r in res:
                for _try in range(64):
                    a = int(rng.integers(0, 360))
                    x = a * q + int(r)
                    if any((x - y) % M in (0, 1, M - 1) for y in addrs):
                        continue        # duplicate row, or adjacent rows (4-cycle with the staircase)
                    if _makes_4cycle(placed, g, int(r), a, addrs, q):
                        continue
                    addrs.append(x)
                    break
                else:
                    ok = False
                    break
            if not ok:
                break
            for x in addrs:
                placed.setdefault(x % q, []).append((g, x // q))
            table.append(addrs)
        if ok:
            return table
    raise RuntimeError("could not build a DVB-S2-profile table")


def _makes_4cycle(placed, g, r, a, addrs, q):
    """Length-4 cycle test on the 360-circulant structure: entries (g,r,a) and
    (g,r2,a2) of the same group together with entries (h,r,b), (h,r2,b2) of another
    group close a 4-cycle iff a - a2 == b - b2 (mod 360)."""
    for y in addrs:
        r2, a2 = y % q, y // q
        if r2 == r:
            continue
        d = (a - a2) % 360
        others_r2 = {}
        for (h, b2) in placed.get(r2, ()):
            others_r2.setdefault(h, []).append(b2)
        for (h, b) in placed.get(r, ()):
            for b2 in others_r2.get(h, ()):
                if (b - b2) % 360 == d:
                    return True
    return False


def dvbs2_profile_edges(N=64800, K=32400, seed=20260101, profile=None):
    """Row-major edge list of the DVB-S2-profile code: information part from the
    generated address table, parity part the standard's staircase (column K+m has
    ones in rows m and m+1)."""
    table = dvbs2_profile_table(N, K, seed, profile)
    M = N - K
    q = M // 360
    s = np.arange(360, dtype=np.int64)
    rows, cols = [], []
    for g, addrs in enumerate(table):
        for x in addrs:
            rows.append((x + s * q) % M)
            cols.append(360 * g + s)
    m = np.arange(M, dtype=np.int64)
    rows += [m, m[1:]]
    cols += [K + m, K + m[:-1]]
    return row_major(np.concatenate(rows), np.concatenate(cols))


# ---------------------------------------------------------------------------- 5G NR

#: row weights of 3GPP TS 38.212 base graph 1 (46 rows, 316 blocks in total)
BG1_ROW_WEIGHTS = [19, 19, 19, 19, 3, 8, 9, 7, 10, 9, 7, 8, 7, 6, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 6, 5,
                   5, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 4, 5, 4, 5, 5, 4]


def nr_bg1_profile_base(seed=20260101, Z=384):
    """46 x 68 base matrix (-1 = empty, else circulant shift) with BG1's shape: 22
    information + 4 core parity block columns (dual diagonal with a weight-3 column),
    42 extension rows each owning one identity parity column, BG1's row weights."""
    rng = np.random.default_rng(seed)
    base = -np.ones((46, 68), np.int64)
    # core parity: column 22 in rows 0,1,3 (shifts 1,0,1); dual diagonal 23..25
    base[0, 22], base[1, 22], base[3, 22] = 1 % Z, 0, 1 % Z
    base[0, 23] = base[1, 23] = 0
    base[1, 24] = base[2, 24] = 0
    base[2, 25] = base[3, 25] = 0
    for r in range(46):
        have = int((base[r] >= 0).sum())
        if r >= 4:
            base[r, 26 + r - 4] = 0            # own identity parity column
            have += 1
        need = BG1_ROW_WEIGHTS[r] - have
        pool = np.arange(22) if r < 4 else np.arange(26)
        pool = pool[base[r, pool] < 0]
        # the two leading (high-degree, punctured in NR) columns are used by most rows
        w = np.where(pool < 2, 6.0, 1.0)
        pick = rng.choice(pool, size=need, replace=False, p=w / w.sum())
        base[r, pick] = rng.integers(0, Z, size=need)
    assert int((base >= 0).sum()) == 316
    return base


def qc_edges(base, Z):
    """Expand a base matrix of circulant shifts: block (i, j) with shift p has ones at
    (i*Z + r, j*Z + (r + p) % Z)."""
    base = np.asarray(base, np.int64)
    r = np.arange(Z, dtype=np.int64)
    rows, cols = [], []
    for i, j in zip(*np.nonzero(base >= 0)):
        rows.append(i * Z + r)
        cols.append(j * Z + (r + base[i, j]) % Z)
    return row_major(np.concatenate(rows), np.concatenate(cols))


def nr_bg1_profile_edges(Z=384, seed=20260101):
    """Row-major edges of the BG1-profile code: M = 46 Z, N = 68 Z, K = 22 Z,
    E = 316 Z; layers of Z rows have pairwise disjoint columns."""
    return qc_edges(nr_bg1_profile_base(seed, Z), Z)


def _circ(v, p, Z):
    """(P_p v)[r] = v[(r + p) % Z] for every Z-block of v."""
    return np.roll(v.reshape(-1, Z), -int(p), axis=1).reshape(-1)


def nr_bg1_profile_encode(base, Z, info_bits):
    """Systematic encoding for a base matrix with BG1's parity structure (core: weight-3
    column 22 in rows 0,1,3 with shifts (a,0,a) + dual diagonal 23..25; extension rows own
    an identity column).  info_bits: uint8 [22*Z].  Returns the codeword uint8 [68*Z]."""
    base = np.asarray(base, np.int64)
    cw = np.zeros(68 * Z, np.uint8)
    cw[:22 * Z] = info_bits

    def lam(r, upto):
        acc = np.zeros(Z, np.uint8)
        for j in range(upto):
            if base[r, j] >= 0:
                acc ^= _circ(cw[j * Z:(j + 1) * Z], base[r, j], Z)
        return acc

    l = [lam(r, 22) for r in range(4)]
    a = base[0, 22]
    pa = l[0] ^ l[1] ^ l[2] ^ l[3]
    pb = l[0] ^ _circ(pa, a, Z)
    pc = l[1] ^ pa ^ pb
    pd = l[2] ^ pc
    cw[22 * Z:26 * Z] = np.concatenate([pa, pb, pc, pd])
    for r in range(4, 46):
        cw[(26 + r - 4) * Z:(27 + r - 4) * Z] = lam(r, 26)
    return cw


# ----------------------------------------------------------------------------- alist

def load_alist(path):
    """Read a parity-check matrix in MacKay's alist format (the usual interchange format
    for LDPC matrices) and return (rows, cols, M, N) in row-major edge order.

    Layout: "N M", "max_col_deg max_row_deg", N column degrees, M row degrees, then N lines of
    1-based row indices per column (0-padded) and M lines of 1-based column indices per row.
    Only the per-column lists are used; the per-row lists are checked against them."""
    tok = open(path).read().split()
    it = iter(int(t) for t in tok)
    N, M = next(it), next(it)
    next(it), next(it)
    cdeg = [next(it) for _ in range(N)]
    rdeg = [next(it) for _ in range(M)]
    maxc, maxr = max(cdeg), max(rdeg)
    rows, cols = [], []
    rest = list(it)
    pos = 0
    # each column line holds either exactly deg entries or max_col_deg entries (0-padded)
    padded = len(rest) >= N * maxc + M * maxr
    for c in range(N):
        w = maxc if padded else cdeg[c]
        ent = [r for r in rest[pos:pos + w] if r > 0]
        pos += w
        if len(ent) != cdeg[c]:
            raise ValueError("alist: column %d lists %d rows, header says %d" % (c, len(ent), cdeg[c]))
        rows += [r - 1 for r in ent]
        cols += [c] * len(ent)
    rr, cc = row_major(rows, cols)
    if not np.array_equal(np.bincount(rr, minlength=M), np.array(rdeg)):
        raise ValueError("alist: row degrees do not match the column lists")
    return rr, cc, M, N


def save_alist(path, rows, cols, M, N):
    """Write an edge list as alist (0-padded lines)."""
    rows, cols = np.asarray(rows), np.asarray(cols)
    cdeg, rdeg = np.bincount(cols, minlength=N), np.bincount(rows, minlength=M)
    by_col = [[] for _ in range(N)]
    by_row = [[] for _ in range(M)]
    for r, c in zip(rows.tolist(), cols.tolist()):
        by_col[c].append(r + 1)
        by_row[r].append(c + 1)
    with open(path, "w") as f:
        f.write("%d %d\n%d %d\n" % (N, M, cdeg.max(), rdeg.max()))
        f.write(" ".join(map(str, cdeg)) + "\n" + " ".join(map(str, rdeg)) + "\n")
        for c in range(N):
            f.write(" ".join(map(str, sorted(by_col[c]) + [0] * (cdeg.max() - cdeg[c]))) + "\n")
        for r in range(M):
            f.write(" ".join(map(str, sorted(by_row[r]) + [0] * (rdeg.max() - rdeg[r]))) + "\n")

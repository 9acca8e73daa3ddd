"""The contact cache of one env, decoded from its row form (rp_kernels.cuh PMC_*: 704 words behind the 128-float state record of rp_get_state; the CPU oracle exports
its own cache in the same layout, OracleEnv.get_cache_row).  Shared by the CPU and GPU tests that compare caches field by field.

What a row holds that is STATE: the manifolds in creation order (object-pair key, points in slot order: colliders, the two body-frame points, normal, distance, the
breaking threshold) and the 16 cached GJK results (tag = baked pair index + 1, simplex size, box-core corner codes, hull vertex numbers, direction).  What it also holds and
is NOT state: the pair flags (rebuilt every substep), four scratch words per manifold (slot owners of the matching step), stale words behind a manifold's points and
inside a GJK slot whose tag is 0 - ignored here."""
import numpy as np

HDR, PT, MAN, PM_MAX, AXN = 4, 11, 8 + 4 * 11, 11, 16
AX = HDR + PM_MAX * MAN
WORDS = AX + 8 * AXN


def decode(row):
    """row: float32 [704] -> dict(manifolds=[dict(key, n, thr, ab=[...], lA, lB, nrm, dist)], gjk={slot: dict(tag, n, codes, vi, v)})"""
    row = np.ascontiguousarray(row, dtype=np.float32)
    assert row.shape == (WORDS,), row.shape
    w = row.view(np.int32)
    npm = int(w[0])
    assert 0 <= npm <= PM_MAX, npm
    mans = []
    for i in range(npm):
        b = HDR + MAN * i
        n = int(w[b + 1])
        assert 0 <= n <= 4, n
        p = row[b + 8:b + 8 + PT * 4].reshape(4, PT)[:n]
        pw = w[b + 8:b + 8 + PT * 4].reshape(4, PT)[:n]
        mans.append(dict(key=int(w[b]), n=n, thr=float(row[b + 2]), ab=[int(x) & 0xFFFF for x in pw[:, 10]],
                         lA=p[:, 0:3].astype(np.float64), lB=p[:, 3:6].astype(np.float64), nrm=p[:, 6:9].astype(np.float64), dist=p[:, 9].astype(np.float64)))
    gjk = {}
    for s in range(AXN):
        b = AX + 8 * s
        tag = int(w[b])
        if tag == 0:
            continue
        nc = int(w[b + 1])
        n = min(max(nc & 15, 1), 3)
        gjk[s] = dict(tag=tag, n=n, codes=[(nc >> (4 + 4 * k)) & 7 for k in range(n)], vi=[int(w[b + 2 + k]) for k in range(n)], v=row[b + 5:b + 8].astype(np.float64))
    return dict(manifolds=mans, gjk=gjk)


def integers(row):
    """the discrete content of a cache row, as one comparable tuple: (manifold keys with their points' colliders in slot order, GJK slots with tag / simplex)"""
    d = decode(row)
    return (tuple((m['key'], tuple(m['ab'])) for m in d['manifolds']),
            tuple((s, g['tag'], g['n'], tuple(g['codes']), tuple(g['vi'])) for s, g in sorted(d['gjk'].items())))


def features(row):
    """like integers(), with a GJK slot reduced to the FEATURES its simplex spans: the set of hull vertices and the set of box-core corners.  Two runs that agree on a
    closest point to rounding may hold it as different simplices of the same features (an edge-edge contact's Minkowski face is a parallelogram: either triangulation;
    a plate's core has coincident corners) - equivalent warm starts, the same cached direction"""
    d = decode(row)
    return (tuple((m['key'], tuple(m['ab'])) for m in d['manifolds']),
            tuple((s, g['tag'], frozenset(g['vi']), frozenset(g['codes'])) for s, g in sorted(d['gjk'].items())))


def manifolds(row):
    """the manifold half of integers(): keys in creation order, each with its points' colliders in slot order - what decides the solver's rows"""
    return integers(row)[0]


def gjk_tags(row):
    """which (slot, pair) entries the GJK cache holds"""
    return tuple((g[0], g[1]) for g in integers(row)[1])


def float_gap(row_a, row_b):
    """largest difference of the continuous content of two rows whose discrete content agrees: body-frame points and distances [m], normals and GJK directions
    (relative to the direction's length)"""
    a, b = decode(row_a), decode(row_b)
    gap = 0.0
    for ma, mb in zip(a['manifolds'], b['manifolds']):
        if ma['n']:
            gap = max(gap, float(np.abs(ma['lA'] - mb['lA']).max()), float(np.abs(ma['lB'] - mb['lB']).max()), float(np.abs(ma['nrm'] - mb['nrm']).max()),
                      float(np.abs(ma['dist'] - mb['dist']).max()))
    for s in a['gjk']:
        va, vb = a['gjk'][s]['v'], b['gjk'][s]['v']
        gap = max(gap, float(np.abs(va - vb).max() / max(1e-9, np.abs(va).max())))
    return gap


def describe(row):
    d = decode(row)
    return 'manifolds %s | gjk %s' % (['%d:%s' % (m['key'], ['%d-%d' % (x & 255, x >> 8) for x in m['ab']]) for m in d['manifolds']],
                                      ['s%d pair %d n%d vi%s c%s' % (s, g['tag'] - 1, g['n'], g['vi'], g['codes']) for s, g in sorted(d['gjk'].items())])

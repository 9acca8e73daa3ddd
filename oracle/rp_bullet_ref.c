/* oracle/rp_bullet_ref.c — the FROZEN Bullet-like reference step ("mode B").  TEST INFRASTRUCTURE ONLY.
 *
 * Included by rp_oracle.c when it is compiled with -DRPO_BULLET_REF (librp_oracle_bullet.so); it replaces rpo_substep's
 * collision and constraint solve and keeps everything else (harness, ABA, unconstrained velocities, integration).
 *
 * WHY IT EXISTS.  rp_oracle.c's own physics ("mode A", the fast model) is the model the HIP kernels implement: stateless
 * contact points rebuilt every substep, every collider a box or a sphere, a fixed row order, one application point per
 * contact.  Those choices were made with the kernels in mind.  This file restates what is recalled of Bullet's own step
 * WITHOUT regard for the kernels, so that the fast model can be measured against it (tools/model_divergence.py, DESIGN.md
 * section 2).  Kernel work must not touch this file; it changes only when a recollection of Bullet is corrected.
 *
 * PARITY UNPINNED like the rest of the physics: PyBullet is absent, every block cites the upstream bullet3 source it restates
 * FROM MEMORY (file and function names of bullet3 2.8x / 3.x; none of that source is in /root/reference).
 *
 * What differs from mode A (one flag each, rpo_set_ref_flags, so that every difference can be switched on its own):
 *   RPB_HULL     arm colliders are what the URDF importer builds: btConvexHullShape of the collision mesh (margin 0.001),
 *                btCylinderShapeZ; they collide through GJK / EPA (btGjkPairDetector, btGjkEpa2), one point per step, like
 *                every pair that is not box-box (BulletUrdfImporter::convertURDFToCollisionShape, btConvexConvexAlgorithm)
 *   RPB_PERSIST  persistent manifolds (btPersistentManifold): <= 4 points per manifold, points live in the two bodies' local
 *                frames, are refreshed every step and removed beyond the manifold's contact breaking threshold (normal and
 *                tangential drift); box-box (btBoxBoxDetector = ODE dBoxBox2) adds points only while the boxes overlap;
 *                a new point within the threshold of a cached one replaces it.  No cap on the number of manifolds.
 *   RPB_ORDER    btMultiBodyConstraintSolver::solveSingleIteration's order: non-contact rows (limits, motors, gear) in creation
 *                order and in ALTERNATING direction from sweep to sweep, then normals, torsional friction, friction
 *   RPB_LIMIT    a joint-limit row exists only while the limit is violated, and pushes back with erp 0.2
 *                (btMultiBodyJointLimitConstraint::createConstraintRows: `if (penetration > 0) continue;`, m_erp); mode A keeps
 *                a speculative row from 0.1 rad before the limit on, which stops the joint exactly at the limit
 *   RPB_LEVER    a contact acts at positionWorldOnA on body A and positionWorldOnB on body B (mode A: their midpoint on both)
 *   RPB_SOFT     <contact> stiffness / damping of the gripper links (ur5e2.urdf:306-312, panda.urdf:256-262) become the
 *                row's cfm and erp (BT_CONTACT_FLAG_CONTACT_STIFFNESS_DAMPING in setupMultiBodyContactConstraint)
 *   RPB_ANCHOR   friction_anchor: friction rows pull the two anchor points of a persistent contact together (frictionERP 0.2)
 *   RPB_SPIN     spinning_friction: one torsional friction row per manifold
 *   RPB_FRICSKIP a friction row is skipped while its normal impulse is zero (its impulse is kept, not clamped to 0)
 *   RPB_WARM     warm starting of the normal rows from the cached impulse (factor 0.85).  OFF in RPB_DEFAULT: recollection
 *                of btMultiBodyConstraintSolver::setupMultiBodyContactConstraint is that warm starting of multibody contact
 *                rows is disabled in the sources of that era ("issues/bugs in the warmstarting", bullet3 issue 1476)
 * Not restated: convex-vs-triangle-mesh collision for the drawer and the door (they keep mode A's exact box decomposition of
 * the meshes, each body sharing ONE manifold per partner as btConvexConcaveCollisionAlgorithm does), implicit cone friction,
 * island splitting / constraint batching (row order across bodies), rolling friction (all zero here). */
#ifndef RPO_BULLET_REF
#error "rp_bullet_ref.c is included by rp_oracle.c under -DRPO_BULLET_REF"
#endif
#ifdef RP_FLOAT
#error "the Bullet-like reference is built in fp64 only"
#endif
#include "generated/rp_hulls_gen.h"

#define RPB_HULL 1
#define RPB_PERSIST 2
#define RPB_ORDER 4
#define RPB_LEVER 8
#define RPB_SOFT 16
#define RPB_ANCHOR 32
#define RPB_SPIN 64
#define RPB_FRICSKIP 128
#define RPB_WARM 256
#define RPB_LIMIT 512
#define RPB_DEFAULT (RPB_HULL | RPB_PERSIST | RPB_ORDER | RPB_LEVER | RPB_SOFT | RPB_ANCHOR | RPB_SPIN | RPB_FRICSKIP | RPB_LIMIT)

#define RPB_SHAPE_MARGIN 0.001          /* gUrdfDefaultCollisionMargin / the physics server's default collision margin */
#define RPB_BREAKING 0.02               /* gContactBreakingThreshold; a manifold's threshold is relative (rp_model.col_thr) */
#define RPB_ERP_LIMIT 0.2               /* btContactSolverInfo::m_erp */
#define RPB_FRICTION_ERP 0.2            /* m_frictionERP */
#define RPB_WARM_FACTOR 0.85            /* m_warmstartingFactor default */
#define RPB_MAX_MAN 160
#define RPB_MAX_CON (4 * RPB_MAX_MAN)
#define RPB_MAX_ROWS (RP_MAX_ARM * 3 + RP_MAX_J1 + 2 + 4 * RPB_MAX_CON)

typedef struct {
  real lA[3], lB[3];        /* btManifoldPoint::m_localPointA / B (here: in the frames of the two BODIES) */
  real n[3];                /* m_normalWorldOnB: from B toward A */
  real pA[3], pB[3], dist;  /* refreshed world points, distance1 */
  real imp, imp1, imp2;     /* applied impulses of the previous step (warm starting) */
  int life;
} rpb_point;

typedef struct {
  int key_a, key_b;         /* collider pair, or object pair for the bodies that share one manifold (drawer, door) */
  int ca, cb;               /* colliders the manifold was created for (bodies, friction, contact block) */
  int n, touched;
  real thr;
  rpb_point p[4];
} rpb_manifold;

typedef struct { real J[RP_MAX_NV], B[RP_MAX_NV], rhs, lo, hi, dinv, cfm, lambda, mu; int parent; } rpb_row;

typedef struct {
  unsigned flags;
  int nman;
  rpb_manifold man[RPB_MAX_MAN];
  const rpb_shape* shape[RP_MAX_COL];   /* exact shape / contact block of a collider, NULL = the box or sphere of rp_model */
  int nrows, n_noncontact, n_normal, n_tors, n_fric, ncon;
  rpb_row* rows;
  rpb_point* cpt[RPB_MAX_CON];          /* contact of normal row k */
  int overflow;
} rpb_state;

/* ------------------------------------------------------------------ shapes and support mappings (cores, without margin) */
typedef struct { int kind; const real* R; const real* p; real he[3]; real radius, halflen; int n; const double* v; real margin; } rpb_cvx;
/* kind 0 box, 1 sphere, 2 hull (vertices in the frame R, p), 3 cylinder along local z */

static void rpb_support(const rpb_cvx* s, const real* dir, real* out) {
  real dl[3], l[3];
  m3tmulv(dl, s->R, dir);
  if (s->kind == 0) {                         /* btBoxShape::localGetSupportingVertexWithoutMargin */
    for (int k = 0; k < 3; k++) l[k] = dl[k] >= 0 ? s->he[k] : -s->he[k];
  } else if (s->kind == 1) {                  /* btSphereShape: a point, the radius is its margin */
    l[0] = l[1] = l[2] = 0;
  } else if (s->kind == 2) {                  /* btConvexHullShape: the vertex of largest projection */
    int best = 0; real bd = -1e300;
    for (int i = 0; i < s->n; i++) {
      real d = s->v[3 * i] * dl[0] + s->v[3 * i + 1] * dl[1] + s->v[3 * i + 2] * dl[2];
      if (d > bd) { bd = d; best = i; }
    }
    l[0] = s->v[3 * best]; l[1] = s->v[3 * best + 1]; l[2] = s->v[3 * best + 2];
  } else {                                    /* btCylinderShapeZ (CylinderLocalSupportZ) */
    real sxy = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
    if (sxy > 1e-12) { l[0] = dl[0] * s->radius / sxy; l[1] = dl[1] * s->radius / sxy; } else { l[0] = s->radius; l[1] = 0; }
    l[2] = dl[2] < 0 ? -s->halflen : s->halflen;
  }
  m3mulv(out, s->R, l);
  v3add(out, out, s->p);
}

/* ------------------------------------------------------------------ GJK distance between two cores (btGjkPairDetector + btVoronoiSimplexSolver)
 * Simplex of Minkowski-difference points w = a - b with the support points kept for the witnesses.  Returns 1 and the closest
 * points when the cores are apart, 0 when they touch or overlap (EPA then works on the shapes with their margins). */
typedef struct { real w[3], a[3], b[3]; } rpb_sv;

static void rpb_closest_on_simplex(rpb_sv* s, int* n, real* lam) {
  /* closest point to the origin on the simplex s[0..n); reduces the simplex to the supporting sub-simplex, lam = barycentric */
  if (*n == 1) { lam[0] = 1; return; }
  if (*n == 2) {
    real ab[3]; v3sub(ab, s[1].w, s[0].w);
    real t = -v3dot(s[0].w, ab), den = v3dot(ab, ab);
    if (t <= 0 || den <= 0) { *n = 1; lam[0] = 1; return; }
    if (t >= den) { s[0] = s[1]; *n = 1; lam[0] = 1; return; }
    lam[1] = t / den; lam[0] = 1 - lam[1];
    return;
  }
  if (*n == 3) {                                /* Ericson, closest point on triangle, to the origin */
    const real *a = s[0].w, *b = s[1].w, *c = s[2].w;
    real ab[3], ac[3], ap[3], bp[3], cp[3];
    v3sub(ab, b, a); v3sub(ac, c, a);
    v3scale(ap, a, -1); v3scale(bp, b, -1); v3scale(cp, c, -1);
    real d1 = v3dot(ab, ap), d2 = v3dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { *n = 1; lam[0] = 1; return; }
    real d3 = v3dot(ab, bp), d4 = v3dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { s[0] = s[1]; *n = 1; lam[0] = 1; return; }
    real vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { real v = d1 / (d1 - d3); *n = 2; lam[0] = 1 - v; lam[1] = v; return; }
    real d5 = v3dot(ab, cp), d6 = v3dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { s[0] = s[2]; *n = 1; lam[0] = 1; return; }
    real vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { real w = d2 / (d2 - d6); s[1] = s[2]; *n = 2; lam[0] = 1 - w; lam[1] = w; return; }
    real va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { real w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); s[0] = s[1]; s[1] = s[2]; *n = 2; lam[0] = 1 - w; lam[1] = w; return; }
    real den = 1 / (va + vb + vc);
    lam[1] = vb * den; lam[2] = vc * den; lam[0] = 1 - lam[1] - lam[2];
    return;
  }
  /* tetrahedron: the closest of the faces the origin is outside of; inside all = overlap */
  static const int F[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
  static const int OPP[4] = {3, 1, 2, 0};
  real best = 1e300; int bf = -1; rpb_sv bs[3]; int bn = 0; real bl[3] = {0, 0, 0};
  for (int f = 0; f < 4; f++) {
    const real *a = s[F[f][0]].w, *b = s[F[f][1]].w, *c = s[F[f][2]].w, *d = s[OPP[f]].w;
    real ab[3], ac[3], nrm[3], ad[3];
    v3sub(ab, b, a); v3sub(ac, c, a); v3cross(nrm, ab, ac); v3sub(ad, d, a);
    real so = -v3dot(a, nrm), sd = v3dot(ad, nrm);
    if (so * sd > 0) continue;                  /* origin on the same side as the fourth vertex: not outside this face */
    if (sd == 0 && so == 0) continue;
    rpb_sv t[3] = {s[F[f][0]], s[F[f][1]], s[F[f][2]]}; int tn = 3; real tl[3];
    rpb_closest_on_simplex(t, &tn, tl);
    real q[3] = {0, 0, 0};
    for (int i = 0; i < tn; i++) v3axpy(q, tl[i], t[i].w);
    real dd = v3dot(q, q);
    if (dd < best) { best = dd; bf = f; bn = tn; for (int i = 0; i < tn; i++) { bs[i] = t[i]; bl[i] = tl[i]; } }
  }
  if (bf < 0) { *n = 4; lam[0] = lam[1] = lam[2] = lam[3] = 0.25; return; }     /* origin inside */
  for (int i = 0; i < bn; i++) { s[i] = bs[i]; lam[i] = bl[i]; }
  *n = bn;
}

static void rpb_support_m(const rpb_cvx* s, const real* dir, real* out);
static int rpb_gjk_ex(const rpb_cvx* A, const rpb_cvx* Bs, int with_margin, real* pa, real* pb, real* dist_out, rpb_sv* simplex_out, int* ns_out) {
  rpb_sv s[4]; int n = 0; real lam[4];
  real v[3]; v3sub(v, A->p, Bs->p);
  if (v3dot(v, v) < 1e-20) v3set(v, 1, 0, 0);
  real dd = 1e300;
  for (int it = 0; it < 128; it++) {
    real nv[3], w[3]; rpb_sv sv;
    v3scale(nv, v, -1);
    if (with_margin) { rpb_support_m(A, nv, sv.a); rpb_support_m(Bs, v, sv.b); }
    else { rpb_support(A, nv, sv.a); rpb_support(Bs, v, sv.b); }
    v3sub(w, sv.a, sv.b); v3cpy(sv.w, w);
    real vv = v3dot(v, v), vw = v3dot(v, w);
    int dup = 0;
    for (int i = 0; i < n; i++) { real d[3]; v3sub(d, s[i].w, w); if (v3dot(d, d) < 1e-24) dup = 1; }
    if (dup || (n > 0 && vv - vw <= 1e-12 * vv)) break;          /* no progress: v is the closest point */
    s[n++] = sv;
    rpb_closest_on_simplex(s, &n, lam);
    if (n == 4) { if (simplex_out) { for (int i = 0; i < 4; i++) simplex_out[i] = s[i]; *ns_out = 4; } return 0; }
    real q[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) v3axpy(q, lam[i], s[i].w);
    real nd = v3dot(q, q);
    v3cpy(v, q);
    if (nd < 1e-20) { if (simplex_out) { for (int i = 0; i < n; i++) simplex_out[i] = s[i]; *ns_out = n; } return 0; }
    if (nd >= dd * (1 - 1e-14) && it > 0) { dd = nd; break; }
    dd = nd;
  }
  real a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) { v3axpy(a, lam[i], s[i].a); v3axpy(b, lam[i], s[i].b); }
  v3cpy(pa, a); v3cpy(pb, b);
  *dist_out = sqrt(v3dot(v, v));
  return 1;
}

static int rpb_gjk(const rpb_cvx* A, const rpb_cvx* Bs, real* pa, real* pb, real* dist_out, rpb_sv* simplex_out, int* ns_out) {
  return rpb_gjk_ex(A, Bs, 0, pa, pb, dist_out, simplex_out, ns_out);
}

/* ------------------------------------------------------------------ EPA on the shapes WITH margin (btGjkEpaSolver2::Penetration with margins)
 * support of the inflated shape = core support + margin * dir / |dir|.  Returns depth > 0, the outward normal of the closest
 * face of A (-) B, and the witness points. */
static void rpb_support_m(const rpb_cvx* s, const real* dir, real* out) {
  rpb_support(s, dir, out);
  real l = v3norm(dir);
  if (l > 0) v3axpy(out, s->margin / l, dir);
}
typedef struct { int v[3]; real n[3], d; int alive; } rpb_face;
#define RPB_EPA_V 160
#define RPB_EPA_F 640
static int rpb_epa_face(rpb_face* f, const rpb_sv* V, int a, int b, int c) {
  real ab[3], ac[3];
  f->v[0] = a; f->v[1] = b; f->v[2] = c; f->alive = 1;
  v3sub(ab, V[b].w, V[a].w); v3sub(ac, V[c].w, V[a].w); v3cross(f->n, ab, ac);
  real l = v3norm(f->n);
  if (l < 1e-18) { f->d = 0; return 0; }
  v3scale(f->n, f->n, 1 / l);
  f->d = v3dot(f->n, V[a].w);
  if (f->d < 0) { int t = f->v[1]; f->v[1] = f->v[2]; f->v[2] = t; v3scale(f->n, f->n, -1); f->d = -f->d; }
  return 1;
}
static int rpb_epa(const rpb_cvx* A, const rpb_cvx* Bs, real* nrm_out, real* depth_out, real* pa, real* pb) {
  rpb_sv V[RPB_EPA_V]; int nv = 0;
  rpb_face F[RPB_EPA_F]; int nf = 0;
  /* initial tetrahedron: the simplex GJK ends with on the inflated shapes (it encloses the origin, since they overlap) */
  {
    real qa[3], qb[3], d0; rpb_sv sx[4]; int ns = 0;
    if (rpb_gjk_ex(A, Bs, 1, qa, qb, &d0, sx, &ns) || ns != 4) return 0;
    for (int i = 0; i < 4; i++) V[nv++] = sx[i];
  }
  if (!rpb_epa_face(&F[nf++], V, 0, 1, 2) || !rpb_epa_face(&F[nf++], V, 0, 2, 3) || !rpb_epa_face(&F[nf++], V, 0, 3, 1) || !rpb_epa_face(&F[nf++], V, 1, 3, 2)) return 0;
  for (int it = 0; it < 128; it++) {
    int bf = -1; real bd = 1e300;
    for (int i = 0; i < nf; i++) if (F[i].alive && F[i].d < bd) { bd = F[i].d; bf = i; }
    if (bf < 0) return 0;
    rpb_sv sv; real nd[3];
    v3scale(nd, F[bf].n, -1);
    rpb_support_m(A, F[bf].n, sv.a); rpb_support_m(Bs, nd, sv.b); v3sub(sv.w, sv.a, sv.b);
    real ext = v3dot(sv.w, F[bf].n);
    if (ext - bd < 1e-9 || nv >= RPB_EPA_V - 1 || nf + 2 * nv + 8 >= RPB_EPA_F || it == 127) {
      /* converged: barycentric coordinates of the projection of the origin on the face */
      const rpb_sv *a = &V[F[bf].v[0]], *b = &V[F[bf].v[1]], *c = &V[F[bf].v[2]];
      real p[3]; v3scale(p, F[bf].n, bd);
      real v0[3], v1[3], v2[3];
      v3sub(v0, b->w, a->w); v3sub(v1, c->w, a->w); v3sub(v2, p, a->w);
      real d00 = v3dot(v0, v0), d01 = v3dot(v0, v1), d11 = v3dot(v1, v1), d20 = v3dot(v2, v0), d21 = v3dot(v2, v1);
      real den = d00 * d11 - d01 * d01;
      real bv = den != 0 ? (d11 * d20 - d01 * d21) / den : 0, bw = den != 0 ? (d00 * d21 - d01 * d20) / den : 0, bu = 1 - bv - bw;
      for (int k = 0; k < 3; k++) { pa[k] = bu * a->a[k] + bv * b->a[k] + bw * c->a[k]; pb[k] = bu * a->b[k] + bv * b->b[k] + bw * c->b[k]; }
      v3cpy(nrm_out, F[bf].n); *depth_out = bd;
      return 1;
    }
    /* remove the faces the new vertex sees, keep the horizon edges */
    int E[RPB_EPA_F][2], ne = 0;
    V[nv] = sv;
    for (int i = 0; i < nf; i++) {
      if (!F[i].alive) continue;
      real t[3]; v3sub(t, sv.w, V[F[i].v[0]].w);
      if (v3dot(F[i].n, t) <= 1e-12) continue;
      F[i].alive = 0;
      for (int e2 = 0; e2 < 3; e2++) {
        int x = F[i].v[e2], y = F[i].v[(e2 + 1) % 3], found = -1;
        for (int k = 0; k < ne; k++) if (E[k][0] == y && E[k][1] == x) found = k;
        if (found >= 0) { E[found][0] = E[ne - 1][0]; E[found][1] = E[ne - 1][1]; ne--; }
        else { E[ne][0] = x; E[ne][1] = y; ne++; }
      }
    }
    if (ne == 0) return 0;
    for (int k = 0; k < ne; k++) {
      int slot = -1;
      for (int i = 0; i < nf; i++) if (!F[i].alive) { slot = i; break; }
      if (slot < 0) slot = nf++;
      rpb_epa_face(&F[slot], V, E[k][0], E[k][1], nv);
    }
    nv++;
  }
  return 0;
}

/* ------------------------------------------------------------------ btBoxBoxDetector (ODE dBoxBox2): contacts only while the boxes overlap.
 * Output: normal from B toward A, points on B (pb) with depth >= 0; at most 4 (cullPoints2 keeps the deepest and spreads the
 * others by angle around the centroid). */
static void rpb_cull_points(int n, const real (*p)[2], int m, int i0, int* iret) {
  real cx, cy;
  if (n == 1) { cx = p[0][0]; cy = p[0][1]; }
  else if (n == 2) { cx = (real)0.5 * (p[0][0] + p[1][0]); cy = (real)0.5 * (p[0][1] + p[1][1]); }
  else {
    real a = 0; cx = 0; cy = 0;
    for (int i = 0; i < n - 1; i++) {
      real q = p[i][0] * p[i + 1][1] - p[i + 1][0] * p[i][1];
      a += q; cx += q * (p[i][0] + p[i + 1][0]); cy += q * (p[i][1] + p[i + 1][1]);
    }
    real q = p[n - 1][0] * p[0][1] - p[0][0] * p[n - 1][1];
    a = fabs(a + q) > 1e-30 ? 1 / (3 * (a + q)) : 1e30;
    cx = a * (cx + q * (p[n - 1][0] + p[0][0])); cy = a * (cy + q * (p[n - 1][1] + p[0][1]));
  }
  real A[8]; int avail[8];
  for (int i = 0; i < n; i++) { A[i] = atan2(p[i][1] - cy, p[i][0] - cx); avail[i] = 1; }
  avail[i0] = 0; iret[0] = i0;
  for (int j = 1; j < m; j++) {
    real a = j * (2 * M_PI / m) + A[i0];
    if (a > M_PI) a -= 2 * M_PI;
    real maxdiff = 1e9; int pick = i0;
    for (int i = 0; i < n; i++) if (avail[i]) {
      real diff = fabs(A[i] - a);
      if (diff > M_PI) diff = 2 * M_PI - diff;
      if (diff < maxdiff) { maxdiff = diff; pick = i; }
    }
    avail[pick] = 0; iret[j] = pick;
  }
}

typedef struct { real pb[3], depth; } rpb_bbpt;
static int rpb_box_box(const real* ca, const real* Ra, const real* ha, const real* cb, const real* Rb, const real* hb, real* normal_out, rpb_bbpt* out) {
  real A[3][3], Bx[3][3], pp[3], t[3];
  for (int i = 0; i < 3; i++) { col_axis(A[i], Ra, i); col_axis(Bx[i], Rb, i); }
  v3sub(t, cb, ca);                                        /* p = centre of B relative to A, in world and in A's frame */
  for (int i = 0; i < 3; i++) pp[i] = v3dot(A[i], t);
  real R[3][3], Q[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = v3dot(A[i], Bx[j]); Q[i][j] = fabs(R[i][j]) + 1e-5; }   /* fudge2 */
  real s = -1e300, nR[3] = {0, 0, 0}; int code = 0, invert = 0;
  /* face axes of A, then of B */
  for (int i = 0; i < 3; i++) {
    real s2 = fabs(pp[i]) - (ha[i] + hb[0] * Q[i][0] + hb[1] * Q[i][1] + hb[2] * Q[i][2]);
    if (s2 > 0) return 0;
    if (s2 > s) { s = s2; v3cpy(nR, A[i]); invert = pp[i] < 0; code = i + 1; }
  }
  for (int j = 0; j < 3; j++) {
    real e = v3dot(Bx[j], t);
    real s2 = fabs(e) - (ha[0] * Q[0][j] + ha[1] * Q[1][j] + ha[2] * Q[2][j] + hb[j]);
    if (s2 > 0) return 0;
    if (s2 > s) { s = s2; v3cpy(nR, Bx[j]); invert = e < 0; code = j + 4; }
  }
  /* edge axes A_i x B_j, accepted only if clearly better (fudge factor 1.05) */
  real nC[3] = {0, 0, 0}; int edge = 0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    real expr1 = pp[i2] * R[i1][j] - pp[i1] * R[i2][j];
    real expr2 = ha[i1] * Q[i2][j] + ha[i2] * Q[i1][j] + hb[j1] * Q[i][j2] + hb[j2] * Q[i][j1];
    real s2 = fabs(expr1) - expr2;
    if (s2 > 2.2e-16) return 0;
    real n3[3] = {0, 0, 0};                                /* the axis in A's frame: e_i x (column j of R) */
    n3[i1] = -R[i2][j]; n3[i2] = R[i1][j];
    real l = sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
    if (l > 2.2e-16) {
      s2 /= l;
      if (s2 * 1.05 > s) { s = s2; edge = 1; for (int k = 0; k < 3; k++) nC[k] = n3[k] / l; invert = expr1 < 0; code = 7 + 3 * i + j; }
    }
  }
  if (!code) return 0;
  real normal[3];                                          /* from A to B, ODE's convention */
  if (edge && code >= 7) { normal[0] = normal[1] = normal[2] = 0; for (int k = 0; k < 3; k++) v3axpy(normal, nC[k], A[k]); }
  else v3cpy(normal, nR);
  if (invert) v3scale(normal, normal, -1);
  real depth = -s;
  v3scale(normal_out, normal, -1);                         /* m_normalWorldOnB: from B toward A */
  if (code >= 7) {                                         /* edge against edge: one point, reported on B */
    real pa[3], pb[3];
    v3cpy(pa, ca);
    for (int j = 0; j < 3; j++) v3axpy(pa, (v3dot(normal, A[j]) > 0 ? 1 : -1) * ha[j], A[j]);
    v3cpy(pb, cb);
    for (int j = 0; j < 3; j++) v3axpy(pb, (v3dot(normal, Bx[j]) > 0 ? -1 : 1) * hb[j], Bx[j]);
    const real* ua = A[(code - 7) / 3];
    const real* ub = Bx[(code - 7) % 3];
    real d[3]; v3sub(d, pb, pa);
    real uaub = v3dot(ua, ub), q1 = v3dot(ua, d), q2 = -v3dot(ub, d), dd = 1 - uaub * uaub, alpha = 0, beta = 0;
    if (dd > 1e-4) { dd = 1 / dd; alpha = (q1 + uaub * q2) * dd; beta = (uaub * q1 + q2) * dd; }
    v3axpy(pb, beta, ub);
    v3cpy(out[0].pb, pb); out[0].depth = depth;
    return 1;
  }
  /* face contact: reference box (a), incident box (b) */
  const real *Ra_[3], *Rb_[3], *pa_, *pb_, *Sa, *Sb;
  if (code <= 3) { for (int k = 0; k < 3; k++) { Ra_[k] = A[k]; Rb_[k] = Bx[k]; } pa_ = ca; pb_ = cb; Sa = ha; Sb = hb; }
  else { for (int k = 0; k < 3; k++) { Ra_[k] = Bx[k]; Rb_[k] = A[k]; } pa_ = cb; pb_ = ca; Sa = hb; Sb = ha; }
  real normal2[3], nr[3], anr[3];
  if (code <= 3) v3cpy(normal2, normal); else v3scale(normal2, normal, -1);
  for (int k = 0; k < 3; k++) { nr[k] = v3dot(Rb_[k], normal2); anr[k] = fabs(nr[k]); }
  int lanr, a1, a2;
  if (anr[1] > anr[0]) { if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  else { if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  real center[3];
  for (int k = 0; k < 3; k++) center[k] = pb_[k] - pa_[k] + (nr[lanr] < 0 ? 1 : -1) * Sb[lanr] * Rb_[lanr][k];
  int codeN = code <= 3 ? code - 1 : code - 4, code1, code2;
  if (codeN == 0) { code1 = 1; code2 = 2; } else if (codeN == 1) { code1 = 0; code2 = 2; } else { code1 = 0; code2 = 1; }
  real quad[4][2], c1 = v3dot(center, Ra_[code1]), c2 = v3dot(center, Ra_[code2]);
  real m11 = v3dot(Ra_[code1], Rb_[a1]), m12 = v3dot(Ra_[code1], Rb_[a2]), m21 = v3dot(Ra_[code2], Rb_[a1]), m22 = v3dot(Ra_[code2], Rb_[a2]);
  {
    real k1 = m11 * Sb[a1], k2 = m21 * Sb[a1], k3 = m12 * Sb[a2], k4 = m22 * Sb[a2];
    quad[0][0] = c1 - k1 - k3; quad[0][1] = c2 - k2 - k4;
    quad[1][0] = c1 - k1 + k3; quad[1][1] = c2 - k2 + k4;
    quad[2][0] = c1 + k1 + k3; quad[2][1] = c2 + k2 + k4;
    quad[3][0] = c1 + k1 - k3; quad[3][1] = c2 + k2 - k4;
  }
  real rect[2] = {Sa[code1], Sa[code2]};
  /* intersectRectQuad2: clip the quad against the four sides of the rectangle */
  real buf[2][8][2]; int nq = 4, cur = 0;
  for (int i = 0; i < 4; i++) { buf[0][i][0] = quad[i][0]; buf[0][i][1] = quad[i][1]; }
  for (int dir = 0; dir <= 1; dir++)
    for (int sign = -1; sign <= 1; sign += 2) {
      int nr2 = 0;
      for (int i = 0; i < nq; i++) {
        const real* pq = buf[cur][i];
        const real* nx = buf[cur][(i + 1) % nq];
        if (sign * pq[dir] < rect[dir]) { buf[cur ^ 1][nr2][0] = pq[0]; buf[cur ^ 1][nr2][1] = pq[1]; nr2++; if (nr2 == 8) goto clipped; }
        if ((sign * pq[dir] < rect[dir]) ^ (sign * nx[dir] < rect[dir])) {
          buf[cur ^ 1][nr2][1 - dir] = pq[1 - dir] + (nx[1 - dir] - pq[1 - dir]) / (nx[dir] - pq[dir]) * (sign * rect[dir] - pq[dir]);
          buf[cur ^ 1][nr2][dir] = sign * rect[dir];
          nr2++; if (nr2 == 8) goto clipped;
        }
      }
      nq = nr2; cur ^= 1;
      continue;
    clipped:
      nq = nr2; cur ^= 1;
    }
  if (nq < 1) return 0;
  real det1 = 1 / (m11 * m22 - m12 * m21);
  real point[8][3], dep[8], ret2[8][2]; int cnum = 0;
  for (int j = 0; j < nq; j++) {
    real k1 = (m22 * (buf[cur][j][0] - c1) - m12 * (buf[cur][j][1] - c2)) * det1;
    real k2 = (-m21 * (buf[cur][j][0] - c1) + m11 * (buf[cur][j][1] - c2)) * det1;
    for (int k = 0; k < 3; k++) point[cnum][k] = center[k] + k1 * Rb_[a1][k] + k2 * Rb_[a2][k];
    dep[cnum] = Sa[codeN] - v3dot(normal2, point[cnum]);
    if (dep[cnum] >= 0) { ret2[cnum][0] = buf[cur][j][0]; ret2[cnum][1] = buf[cur][j][1]; cnum++; }
  }
  if (cnum < 1) return 0;
  int keep[8], nk = cnum;
  if (cnum <= 4) { for (int i = 0; i < cnum; i++) keep[i] = i; }
  else {
    int i1 = 0; real maxd = dep[0];
    for (int i = 1; i < cnum; i++) if (dep[i] > maxd) { maxd = dep[i]; i1 = i; }
    rpb_cull_points(cnum, (const real (*)[2])ret2, 4, i1, keep);
    nk = 4;
  }
  for (int j = 0; j < nk; j++) {
    int i = keep[j];
    real pw[3]; v3add(pw, point[i], pa_);                  /* on the incident face */
    if (code <= 3) v3cpy(out[j].pb, pw);                   /* incident box is B */
    else { v3cpy(out[j].pb, pw); v3axpy(out[j].pb, -dep[i], normal); }   /* incident box is A: B's point lies depth along the normal */
    out[j].depth = dep[i];
  }
  return nk;
}

/* ------------------------------------------------------------------ state, shapes of a collider */
static rpb_state* rpb_get(rpo_env* e) {
  if (e->ref) return (rpb_state*)e->ref;
  rpb_state* st = (rpb_state*)calloc(1, sizeof(rpb_state));
  st->flags = RPB_DEFAULT;
  st->rows = (rpb_row*)calloc(RPB_MAX_ROWS, sizeof(rpb_row));
  for (int i = 0; i < rpb_n_shapes; i++)
    if (rpb_shapes[i].kind == e->m.kind && rpb_shapes[i].col < RP_MAX_COL) st->shape[rpb_shapes[i].col] = &rpb_shapes[i];
  e->ref = st;
  return st;
}
void rpo_ref_free(rpo_env* e) { if (e->ref) { free(((rpb_state*)e->ref)->rows); free(e->ref); e->ref = 0; } }
void rpo_set_ref_flags(rpo_env* e, unsigned flags) { rpb_state* st = rpb_get(e); st->flags = flags; st->nman = 0; }
unsigned rpo_get_ref_flags(rpo_env* e) { return rpb_get(e)->flags; }
int rpo_ref_num_contacts(rpo_env* e) { return rpb_get(e)->ncon; }
int rpo_ref_num_manifolds(rpo_env* e) { return rpb_get(e)->nman; }

static void rpb_convex_of(const rpo_env* e, const rpb_state* st, int c, rpb_cvx* s) {
  const rp_model* m = &e->m;
  const rpb_shape* sh = (st->flags & RPB_HULL) ? st->shape[c] : 0;
  memset(s, 0, sizeof(*s));
  s->margin = RPB_SHAPE_MARGIN;
  if (sh && sh->shape == 2) { s->kind = 2; s->R = e->xb[m->col_body[c]].R; s->p = e->xb[m->col_body[c]].p; s->n = sh->n; s->v = sh->v; return; }
  s->R = e->xc[c].R; s->p = e->xc[c].p;
  if (sh && sh->shape == 3) { s->kind = 3; s->radius = sh->radius - RPB_SHAPE_MARGIN; s->halflen = sh->halflen - RPB_SHAPE_MARGIN; return; }
  if (m->col_type[c] == 1) { s->kind = 1; s->margin = m->col_he[c][0]; return; }
  s->kind = 0;
  for (int k = 0; k < 3; k++) { real h = m->col_he[c][k]; real mg = RPB_SHAPE_MARGIN < h ? RPB_SHAPE_MARGIN : h; s->he[k] = h - mg; }
}
static int rpb_is_plain_box(const rpb_state* st, const rp_model* m, int c) {
  const rpb_shape* sh = (st->flags & RPB_HULL) ? st->shape[c] : 0;
  return m->col_type[c] == 0 && !(sh && (sh->shape == 2 || sh->shape == 3));
}

/* world <-> body frame */
static void rpb_to_local(const rpo_env* e, int body, const real* pw, real* pl) { real t[3]; v3sub(t, pw, e->xb[body].p); m3tmulv(pl, e->xb[body].R, t); }
static void rpb_to_world(const rpo_env* e, int body, const real* pl, real* pw) { m3mulv(pw, e->xb[body].R, pl); v3add(pw, pw, e->xb[body].p); }

/* btPersistentManifold::sortCachedPoints: which of the four cached points the new one replaces */
static int rpb_sort_cached(const rpb_manifold* mf, const rpb_point* pt) {
  int maxPenetrationIndex = -1; real maxPenetration = pt->dist;
  for (int i = 0; i < 4; i++) if (mf->p[i].dist < maxPenetration) { maxPenetrationIndex = i; maxPenetration = mf->p[i].dist; }
  real res[4] = {0, 0, 0, 0}, a[3], b[3], cr[3];
  if (maxPenetrationIndex != 0) { v3sub(a, pt->lA, mf->p[1].lA); v3sub(b, mf->p[3].lA, mf->p[2].lA); v3cross(cr, a, b); res[0] = v3dot(cr, cr); }
  if (maxPenetrationIndex != 1) { v3sub(a, pt->lA, mf->p[0].lA); v3sub(b, mf->p[3].lA, mf->p[2].lA); v3cross(cr, a, b); res[1] = v3dot(cr, cr); }
  if (maxPenetrationIndex != 2) { v3sub(a, pt->lA, mf->p[0].lA); v3sub(b, mf->p[3].lA, mf->p[1].lA); v3cross(cr, a, b); res[2] = v3dot(cr, cr); }
  if (maxPenetrationIndex != 3) { v3sub(a, pt->lA, mf->p[0].lA); v3sub(b, mf->p[2].lA, mf->p[1].lA); v3cross(cr, a, b); res[3] = v3dot(cr, cr); }
  int best = 0;                                             /* btVector4::closestAxis4: the largest absolute value */
  for (int i = 1; i < 4; i++) if (fabs(res[i]) > fabs(res[best])) best = i;
  return best;
}

/* btManifoldResult::addContactPoint */
static void rpb_add_point(rpo_env* e, rpb_manifold* mf, const real* nB, const real* pB, real depth_dist) {
  if (depth_dist > mf->thr) return;
  const rp_model* m = &e->m;
  rpb_point pt; memset(&pt, 0, sizeof(pt));
  v3cpy(pt.n, nB); v3cpy(pt.pB, pB); v3cpy(pt.pA, pB); v3axpy(pt.pA, depth_dist, nB); pt.dist = depth_dist;
  rpb_to_local(e, m->col_body[mf->ca], pt.pA, pt.lA);
  rpb_to_local(e, m->col_body[mf->cb], pt.pB, pt.lB);
  int nearest = -1; real shortest = mf->thr * mf->thr;      /* getCacheEntry */
  for (int i = 0; i < mf->n; i++) {
    real d[3]; v3sub(d, mf->p[i].lA, pt.lA);
    real dd = v3dot(d, d);
    if (dd < shortest) { shortest = dd; nearest = i; }
  }
  if (nearest >= 0) {                                       /* replaceContactPoint keeps the cached impulses and the lifetime */
    pt.imp = mf->p[nearest].imp; pt.imp1 = mf->p[nearest].imp1; pt.imp2 = mf->p[nearest].imp2; pt.life = mf->p[nearest].life;
    mf->p[nearest] = pt;
  } else if (mf->n < 4) mf->p[mf->n++] = pt;
  else mf->p[rpb_sort_cached(mf, &pt)] = pt;
}

/* btPersistentManifold::refreshContactPoints */
static void rpb_refresh(rpo_env* e, rpb_manifold* mf) {
  const rp_model* m = &e->m;
  int ba = m->col_body[mf->ca], bb = m->col_body[mf->cb];
  for (int i = mf->n - 1; i >= 0; i--) {
    rpb_point* p = &mf->p[i];
    real d[3];
    rpb_to_world(e, ba, p->lA, p->pA);
    rpb_to_world(e, bb, p->lB, p->pB);
    v3sub(d, p->pA, p->pB);
    p->dist = v3dot(d, p->n);
    p->life++;
  }
  for (int i = mf->n - 1; i >= 0; i--) {
    rpb_point* p = &mf->p[i];
    int drop = 0;
    if (!(p->dist <= mf->thr)) drop = 1;
    else {
      real proj[3], diff[3];
      v3cpy(proj, p->pA); v3axpy(proj, -p->dist, p->n);
      v3sub(diff, p->pB, proj);
      if (v3dot(diff, diff) > mf->thr * mf->thr) drop = 1;
    }
    if (drop) { mf->p[i] = mf->p[mf->n - 1]; mf->n--; }     /* removeContactPoint: the last point takes the slot */
  }
}

static int rpb_shares_manifold(const rp_model* m, int c) {   /* bodies whose colliders are the box decomposition of ONE triangle mesh */
  int b = m->col_body[c];
  int f = b - 1 - m->n_arm;
  if (f >= 0 && f < m->n_free) return m->free_rot_locked[f];
  int j = b - 1 - m->n_arm - m->n_free;
  if (j >= 0 && j < m->n_joint1) { int cnt = 0; for (int k = 0; k < m->n_col; k++) if (m->col_body[k] == b) cnt++; return cnt > 1; }
  return 0;
}

static rpb_manifold* rpb_find_manifold(rpo_env* e, rpb_state* st, int a, int b, int create) {
  const rp_model* m = &e->m;
  int ka = rpb_shares_manifold(m, a) ? 1000 + m->col_obj[a] : a, kb = rpb_shares_manifold(m, b) ? 1000 + m->col_obj[b] : b;
  for (int i = 0; i < st->nman; i++) if (st->man[i].key_a == ka && st->man[i].key_b == kb) return &st->man[i];
  if (!create) return 0;
  if (st->nman >= RPB_MAX_MAN) { st->overflow++; return 0; }
  rpb_manifold* mf = &st->man[st->nman++];
  memset(mf, 0, sizeof(*mf));
  mf->key_a = ka; mf->key_b = kb; mf->ca = a; mf->cb = b;
  /* btCollisionDispatcher::getNewManifold: the smaller of the two objects' relative thresholds */
  mf->thr = m->col_thr[a] < m->col_thr[b] ? m->col_thr[a] : m->col_thr[b];
  return mf;
}

/* collision detection of one step: broadphase over the baked candidate pairs (AABBs grown by gContactBreakingThreshold, as
 * btCollisionWorld::updateSingleAabb does), narrowphase per pair, manifold upkeep */
static void rpb_collide(rpo_env* e, rpb_state* st) {
  const rp_model* m = &e->m;
  const int persist = (st->flags & RPB_PERSIST) != 0;
  if (!persist) st->nman = 0;
  for (int i = 0; i < st->nman; i++) st->man[i].touched = 0;
  for (int pi = 0; pi < m->n_pair; pi++) {
    int a = m->pair[pi][0], b = m->pair[pi][1];
    int sep = 0;
    for (int k = 0; k < 3; k++)
      if (e->aabb_lo[a][k] > e->aabb_hi[b][k] + 2 * RPB_BREAKING + 2 * RPB_SHAPE_MARGIN || e->aabb_lo[b][k] > e->aabb_hi[a][k] + 2 * RPB_BREAKING + 2 * RPB_SHAPE_MARGIN) sep = 1;
    if (sep) continue;
    rpb_manifold* mf = rpb_find_manifold(e, st, a, b, 1);
    if (!mf) continue;
    if (!mf->touched) { mf->touched = 1; }
    /* points of the shared manifolds are stored against the manifold's first collider pair: same two bodies */
    real nB[3];
    if (rpb_is_plain_box(st, m, a) && rpb_is_plain_box(st, m, b)) {
      real ha[3], hb[3]; rpb_bbpt pts[4];
      for (int k = 0; k < 3; k++) { ha[k] = m->col_he[a][k]; hb[k] = m->col_he[b][k]; }
      int np = rpb_box_box(e->xc[a].p, e->xc[a].R, ha, e->xc[b].p, e->xc[b].R, hb, nB, pts);
      for (int i = 0; i < np; i++) rpb_add_point(e, mf, nB, pts[i].pb, -pts[i].depth);
      if (!persist && np == 0) {
        /* without persistence a box pair has no points while it is apart: mode A's stateless points then come from its own
         * margin - here the pair simply has none, which is what Bullet shows on first approach */
      }
    } else {
      rpb_cvx A, Bs; real pa[3], pb[3], dist = 0;
      rpb_convex_of(e, st, a, &A); rpb_convex_of(e, st, b, &Bs);
      rpb_sv simplex[4]; int ns = 0;
      if (rpb_gjk(&A, &Bs, pa, pb, &dist, simplex, &ns)) {
        real v[3]; v3sub(v, pa, pb);
        if (dist > 1e-12) {
          v3scale(nB, v, 1 / dist);
          real d = dist - A.margin - Bs.margin;
          if (d <= mf->thr) {
            real pB[3]; v3cpy(pB, pb); v3axpy(pB, Bs.margin, nB);
            rpb_add_point(e, mf, nB, pB, d);
          }
        }
      } else {
        real nf[3], depth, wa[3], wb[3];
        if (rpb_epa(&A, &Bs, nf, &depth, wa, wb)) {
          v3scale(nB, nf, -1);                              /* the face normal of A (-) B points from A's side: B toward A is its negative */
          rpb_add_point(e, mf, nB, wb, -depth);
        }
      }
    }
  }
  /* refresh after the new points are in (btBoxBoxCollisionAlgorithm / btConvexConvexAlgorithm::processCollision), drop the
   * manifolds whose pair left the broadphase */
  int w = 0;
  for (int i = 0; i < st->nman; i++) {
    if (!st->man[i].touched) continue;
    if (persist) rpb_refresh(e, &st->man[i]);
    if (w != i) st->man[w] = st->man[i];
    w++;
  }
  st->nman = w;
}

/* ------------------------------------------------------------------ rows */
static void rpb_response(const rpo_env* e, int bodyA, const real* pA, int bodyB, const real* pB, const real* dlin, const real* dang, const real* J, real* B) {
  /* B = M^-1 J^T for a row with linear direction dlin at pA on A (+) and at pB on B (-), or a pure angular direction dang */
  const rp_model* m = &e->m;
  memset(B, 0, sizeof(real) * RP_MAX_NV);
  for (int side = 0; side < 2; side++) {
    int body = side == 0 ? bodyA : bodyB;
    const real* p = side == 0 ? pA : pB;
    real sign = side == 0 ? 1 : -1;
    if (body == 0) continue;
    if (body_is_arm(e, body)) {
      real f[6], dqd[RP_MAX_ARM];
      if (dlin) { real sn[3]; v3scale(sn, dlin, sign); v3cross(f, p, sn); v3cpy(f + 3, sn); }
      else { v3scale(f, dang, sign); v3set(f + 3, 0, 0, 0); }
      arm_impulse_response(e, body - 1, f, 0, dqd);
      for (int i = 0; i < m->n_arm; i++) B[i] += dqd[i];
      continue;
    }
    int k = body_free_index(e, body);
    if (k >= 0) {
      int d = dof_free(e, k);
      real im = 1 / (real)m->free_mass[k], w[3];
      for (int i = 0; i < 3; i++) B[d + i] = J[d + i] * im;
      m3mulv(w, e->finv[k], J + d + 3);
      for (int i = 0; i < 3; i++) B[d + 3 + i] = w[i];
      continue;
    }
    k = body_j1_index(e, body);
    if (k >= 0) {
      int d = dof_j1(e, k);
      real minv = m->j1_type[k] == 1 ? 1 / (real)m->j1_mass[k] : 1 / (real)m->j1_inertia_axis[k];
      B[d] = J[d] * minv;
    }
  }
}
static void rpb_ang_jacobian(const rpo_env* e, int body, const real* n, real sign, real* J) {
  const rp_model* m = &e->m;
  if (body == 0) return;
  if (body_is_arm(e, body)) { for (int i = body - 1; i >= 0; i = m->arm_parent[i]) J[i] += sign * v3dot(e->S[i], n); return; }
  int k = body_free_index(e, body);
  if (k >= 0) { int d = dof_free(e, k); for (int i = 0; i < 3; i++) J[d + 3 + i] += sign * n[i]; return; }
  k = body_j1_index(e, body);
  if (k >= 0 && m->j1_type[k] == 0) {
    real ax[3], a[3];
    for (int i = 0; i < 3; i++) ax[i] = (real)m->j1_axis[k][i];
    m3mulv(a, e->xb[body].R, ax);
    J[dof_j1(e, k)] += sign * v3dot(a, n);
  }
}
static rpb_row* rpb_new_row(rpb_state* st) {
  rpb_row* r = &st->rows[st->nrows++];
  memset(r, 0, sizeof(*r));
  r->parent = -1;
  return r;
}

static void rpb_build_rows(rpo_env* e, rpb_state* st, const real* vstar) {
  const rp_model* m = &e->m;
  const int nv = e->nv;
  st->nrows = 0;
  const int order = (st->flags & RPB_ORDER) != 0, blimit = (st->flags & RPB_LIMIT) != 0;
  /* --- non-contact rows.  Creation order in the world: the scene bodies' joint motors (bodies 1, 7, 9 are made before the arm),
   * then the arm - its limit constraints are added while the URDF tree is converted, its motors afterwards - then the gear */
  for (int k = 0; k < m->n_joint1; k++) {
    rpb_row* r = rpb_new_row(st);
    int d = dof_j1(e, k);
    real minv = m->j1_type[k] == 1 ? 1 / (real)m->j1_mass[k] : 1 / (real)m->j1_inertia_axis[k];
    r->J[d] = 1; r->B[d] = minv; r->dinv = 1 / minv;
    real des = 0, maximp = DEFAULT_MOTOR_MAXIMP;
    if (m->j1_has_pos_motor[k]) { des = MOTOR_KP * ((real)m->j1_motor_target[k] - e->jq[k]) / DT; maximp = (real)m->j1_motor_force[k] * DT; }
    r->rhs = (des - vstar[d]) * r->dinv; r->lo = -maximp; r->hi = maximp;
  }
  for (int pass = 0; pass < 2; pass++) {
    const int limits = order ? pass == 0 : pass == 1;       /* mode A's order: motors before limits */
    if (limits) {
      for (int i = 0; i < m->n_arm; i++) {
        if (!(m->arm_lower[i] < m->arm_upper[i])) continue;
        for (int side = 0; side < 2; side++) {
          real pen = side == 0 ? e->q[i] - (real)m->arm_lower[i] : (real)m->arm_upper[i] - e->q[i];
          if (blimit ? pen > 0 : pen > LIMIT_ACTIVATION) continue;     /* btMultiBodyJointLimitConstraint: a row only while violated */
          real sgn = side == 0 ? 1 : -1;
          rpb_row* r = rpb_new_row(st);
          real tau[RP_MAX_ARM] = {0};
          tau[i] = sgn; r->J[i] = sgn;
          arm_impulse_response(e, -1, 0, tau, r->B);
          r->dinv = 1 / (sgn * r->B[i]);
          real relv = sgn * vstar[i], pos_err = 0, vel_err = -relv;
          if (pen > 0) vel_err -= pen / DT; else pos_err = -pen * (blimit ? (real)RPB_ERP_LIMIT : ERP_CONTACT) / DT;
          r->rhs = (pos_err + vel_err) * r->dinv; r->lo = 0; r->hi = LIMIT_MAXIMP;
        }
      }
    } else {
      for (int i = 0; i < m->n_arm; i++) {
        rpb_row* r = rpb_new_row(st);
        real tau[RP_MAX_ARM] = {0};
        tau[i] = 1; r->J[i] = 1;
        arm_impulse_response(e, -1, 0, tau, r->B);
        r->dinv = 1 / r->B[i];
        real des = e->mmode[i] ? MOTOR_KP * (e->mtarget[i] - e->q[i]) / DT : 0;
        r->rhs = (des - vstar[i]) * r->dinv; r->lo = -e->mmaximp[i]; r->hi = e->mmaximp[i];
      }
    }
  }
  if (m->arm_type == RP_ARM_PANDA) {
    int a = dof_of_bullet_joint(e, 9), b = dof_of_bullet_joint(e, 10);
    rpb_row* r = rpb_new_row(st);
    real tau[RP_MAX_ARM] = {0}, ratio = -1;
    tau[a] = 1; tau[b] = ratio; r->J[a] = 1; r->J[b] = ratio;
    arm_impulse_response(e, -1, 0, tau, r->B);
    r->dinv = safe_inv(dotn(r->J, r->B, nv));
    real relv = dotn(r->J, vstar, nv), pos_err = -(e->q[a] + ratio * e->q[b]) * (real)0.1 / DT;
    r->rhs = (pos_err - relv) * r->dinv; r->lo = -(real)50 * DT; r->hi = (real)50 * DT;
  }
  st->n_noncontact = st->nrows;
  /* --- contacts: normals of every manifold point, then one torsional row per manifold, then the friction rows */
  st->ncon = 0;
  const int first_normal = st->nrows;
  for (int mi = 0; mi < st->nman; mi++) {
    rpb_manifold* mf = &st->man[mi];
    int ba = m->col_body[mf->ca], bb = m->col_body[mf->cb];
    for (int i = 0; i < mf->n; i++) {
      if (st->ncon >= RPB_MAX_CON) { st->overflow++; break; }
      rpb_point* p = &mf->p[i];
      real pmid[3], *qa = p->pA, *qb = p->pB;
      if (!(st->flags & RPB_LEVER)) { for (int k = 0; k < 3; k++) pmid[k] = (real)0.5 * (p->pA[k] + p->pB[k]); qa = qb = pmid; }
      rpb_row* r = rpb_new_row(st);
      body_jacobian(e, ba, qa, p->n, 1, r->J);
      body_jacobian(e, bb, qb, p->n, -1, r->J);
      rpb_response(e, ba, qa, bb, qb, p->n, 0, r->J, r->B);
      real d = dotn(r->J, r->B, nv), cfm = 0, erp = ERP_CONTACT;
      const rpb_shape *sa = st->shape[mf->ca], *sb = st->shape[mf->cb];
      if ((st->flags & RPB_SOFT) && ((sa && sa->stiffness > 0) || (sb && sb->stiffness > 0))) {
        /* btManifoldResult: combined stiffness 1 / (1/s0 + 1/s1), combined damping d0 + d1 (defaults 1e18 and 0.1) */
        real s0 = sa && sa->stiffness > 0 ? sa->stiffness : 1e18, s1 = sb && sb->stiffness > 0 ? sb->stiffness : 1e18;
        real d0 = sa && sa->damping >= 0 && sa->stiffness > 0 ? sa->damping : 0.1, d1 = sb && sb->damping >= 0 && sb->stiffness > 0 ? sb->damping : 0.1;
        real ks = 1 / (1 / s0 + 1 / s1), kd = d0 + d1;
        real den = DT * kd + DT * DT * ks;
        cfm = 1 / (den < 2.2e-16 ? 2.2e-16 : den);            /* (a `cfm *= invTimeStep` is recalled after this block: see rp_oracle.c build_rows) */
        erp = (DT * ks) / (DT * ks + kd);
      }
      r->dinv = (d + cfm) > 2.2e-16 ? 1 / (d + cfm) : 0;
      r->cfm = cfm * r->dinv;
      real relv = dotn(r->J, vstar, nv);
      real pen = p->dist + LINEAR_SLOP, pos_err = 0, vel_err = -relv;
      if (pen > 0) vel_err -= pen / DT; else pos_err = -pen * erp / DT;
      r->rhs = (pos_err + vel_err) * r->dinv; r->lo = 0; r->hi = (real)1e10;
      if (st->flags & RPB_WARM) r->lambda = p->imp * (real)RPB_WARM_FACTOR;
      st->cpt[st->ncon++] = p;
    }
  }
  st->n_normal = st->nrows - first_normal;
  const int first_tors = st->nrows;
  if (st->flags & RPB_SPIN) {
    int ci = 0;
    for (int mi = 0; mi < st->nman; mi++) {
      rpb_manifold* mf = &st->man[mi];
      const rpb_shape *sa = st->shape[mf->ca], *sb = st->shape[mf->cb];
      real spin = (sa ? sa->spinning : 0) * (real)m->col_friction[mf->cb] + (sb ? sb->spinning : 0) * (real)m->col_friction[mf->ca];
      if (mf->n > 0 && spin > 0 && ci < st->ncon) {
        int ba = m->col_body[mf->ca], bb = m->col_body[mf->cb];
        rpb_point* p = &mf->p[0];
        rpb_row* r = rpb_new_row(st);
        rpb_ang_jacobian(e, ba, p->n, 1, r->J);
        rpb_ang_jacobian(e, bb, p->n, -1, r->J);
        rpb_response(e, ba, p->pA, bb, p->pB, 0, p->n, r->J, r->B);
        r->dinv = safe_inv(dotn(r->J, r->B, nv));
        r->rhs = -dotn(r->J, vstar, nv) * r->dinv;
        r->parent = first_normal + ci; r->mu = spin;
      }
      ci += mf->n;
    }
  }
  st->n_tors = st->nrows - first_tors;
  {
    int ci = 0;
    for (int mi = 0; mi < st->nman; mi++) {
      rpb_manifold* mf = &st->man[mi];
      int ba = m->col_body[mf->ca], bb = m->col_body[mf->cb];
      const rpb_shape *sa = st->shape[mf->ca], *sb = st->shape[mf->cb];
      int anchor = (st->flags & RPB_ANCHOR) && ((sa && sa->anchor) || (sb && sb->anchor));
      for (int i = 0; i < mf->n && ci < st->ncon; i++, ci++) {
        rpb_point* p = &mf->p[i];
        real t[2][3], pmid[3], *qa = p->pA, *qb = p->pB;
        if (!(st->flags & RPB_LEVER)) { for (int k = 0; k < 3; k++) pmid[k] = (real)0.5 * (p->pA[k] + p->pB[k]); qa = qb = pmid; }
        plane_space(p->n, t[0], t[1]);
        for (int d = 0; d < 2; d++) {
          rpb_row* r = rpb_new_row(st);
          body_jacobian(e, ba, qa, t[d], 1, r->J);
          body_jacobian(e, bb, qb, t[d], -1, r->J);
          rpb_response(e, ba, qa, bb, qb, t[d], 0, r->J, r->B);
          r->dinv = safe_inv(dotn(r->J, r->B, nv));
          real pos_err = 0;
          if (anchor) { real dd[3]; v3sub(dd, p->pA, p->pB); pos_err = -v3dot(dd, t[d]) * (real)RPB_FRICTION_ERP / DT; }
          r->rhs = (pos_err - dotn(r->J, vstar, nv)) * r->dinv;
          r->parent = first_normal + ci;
          r->mu = (real)(m->col_friction[mf->ca] * m->col_friction[mf->cb]);
        }
      }
    }
  }
  st->n_fric = st->nrows - first_tors - st->n_tors;
}

/* btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric */
static void rpb_solve_row(rpb_row* r, real lo, real hi, real* dv, int nv) {
  real delta = r->rhs - r->lambda * r->cfm - dotn(r->J, dv, nv) * r->dinv;
  real sum = r->lambda + delta;
  if (sum < lo) { delta = lo - r->lambda; r->lambda = lo; }
  else if (sum > hi) { delta = hi - r->lambda; r->lambda = hi; }
  else r->lambda = sum;
  for (int i = 0; i < nv; i++) dv[i] += r->B[i] * delta;
}

static void rpb_solve(rpo_env* e, rpb_state* st, real* dv) {
  const int nv = e->nv, nnc = st->n_noncontact;
  if (st->flags & RPB_WARM)        /* the warm-start impulses act before the first sweep */
    for (int k = 0; k < st->n_normal; k++) { rpb_row* r = &st->rows[nnc + k]; for (int i = 0; i < nv; i++) dv[i] += r->B[i] * r->lambda; }
  for (int it = 0; it < N_ITER; it++) {
    for (int j = 0; j < nnc; j++) {
      int index = (st->flags & RPB_ORDER) ? ((it & 1) ? j : nnc - 1 - j) : j;
      rpb_row* r = &st->rows[index];
      rpb_solve_row(r, r->lo, r->hi, dv, nv);
    }
    for (int k = 0; k < st->n_normal; k++) { rpb_row* r = &st->rows[nnc + k]; rpb_solve_row(r, r->lo, r->hi, dv, nv); }
    for (int k = 0; k < st->n_tors + st->n_fric; k++) {
      rpb_row* r = &st->rows[nnc + st->n_normal + k];
      real tot = st->rows[r->parent].lambda;
      if ((st->flags & RPB_FRICSKIP) && !(tot > 0)) continue;
      rpb_solve_row(r, -r->mu * tot, r->mu * tot, dv, nv);
    }
  }
  /* the applied impulses go back into the manifold points (warm starting next step) */
  for (int k = 0; k < st->ncon; k++) st->cpt[k]->imp = st->rows[nnc + k].lambda;
}

/* ------------------------------------------------------------------ test hooks (tests/test_bullet_ref.py) */
/* GJK / EPA between two boxes given as cores he - margin inflated by margin; returns 1 = apart (dist, witness points on the inflated
 * surfaces, normal from B toward A), 2 = overlapping (dist = -depth from EPA), 0 = failure */
int rpo_ref_gjk_epa_boxes(const double* ca, const double* Ra, const double* ha, const double* cb, const double* Rb, const double* hb, double margin,
                          double* dist, double* pa, double* pb, double* nrm) {
  rpb_cvx A, Bs; memset(&A, 0, sizeof(A)); memset(&Bs, 0, sizeof(Bs));
  A.kind = 0; A.R = Ra; A.p = ca; A.margin = margin; Bs.kind = 0; Bs.R = Rb; Bs.p = cb; Bs.margin = margin;
  for (int k = 0; k < 3; k++) { A.he[k] = ha[k] - margin; Bs.he[k] = hb[k] - margin; }
  real qa[3], qb[3], d = 0; rpb_sv sx[4]; int ns = 0;
  if (rpb_gjk(&A, &Bs, qa, qb, &d, sx, &ns) && d > 1e-12) {
    real v[3]; v3sub(v, qa, qb); v3scale(nrm, v, 1 / d);
    *dist = d - 2 * margin;
    for (int k = 0; k < 3; k++) { pa[k] = qa[k] - margin * nrm[k]; pb[k] = qb[k] + margin * nrm[k]; }
    return 1;
  }
  real nf[3], depth;
  if (!rpb_epa(&A, &Bs, nf, &depth, pa, pb)) return 0;
  v3scale(nrm, nf, -1); *dist = -depth;
  return 2;
}
/* btBoxBoxDetector restatement: returns the number of points (<= 4), normal from B toward A, out[4 i .. 4 i + 3] = point on B, depth */
int rpo_ref_box_box(const double* ca, const double* Ra, const double* ha, const double* cb, const double* Rb, const double* hb, double* nrm, double* out) {
  rpb_bbpt pts[4];
  int n = rpb_box_box(ca, Ra, ha, cb, Rb, hb, nrm, pts);
  for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) out[4 * i + k] = pts[i].pb[k]; out[4 * i + 3] = pts[i].depth; }
  return n;
}
/* support-mapped distance between collider a and collider b of the env in its current state (hull / cylinder / box / sphere as the
 * frozen model sees them): returns like rpo_ref_gjk_epa_boxes */
int rpo_ref_collider_distance(rpo_env* e, int a, int b, double* dist, double* pa, double* pb, double* nrm) {
  rpb_state* st = rpb_get(e);
  update_transforms(e);
  rpb_cvx A, Bs; rpb_convex_of(e, st, a, &A); rpb_convex_of(e, st, b, &Bs);
  real qa[3], qb[3], d = 0; rpb_sv sx[4]; int ns = 0;
  if (rpb_gjk(&A, &Bs, qa, qb, &d, sx, &ns) && d > 1e-12) {
    real v[3]; v3sub(v, qa, qb); v3scale(nrm, v, 1 / d);
    *dist = d - A.margin - Bs.margin;
    for (int k = 0; k < 3; k++) { pa[k] = qa[k] - A.margin * nrm[k]; pb[k] = qb[k] + Bs.margin * nrm[k]; }
    return 1;
  }
  real nf[3], depth;
  if (!rpb_epa(&A, &Bs, nf, &depth, pa, pb)) return 0;
  v3scale(nrm, nf, -1); *dist = -depth;
  return 2;
}
/* manifold dump: per point [key_a, key_b, ca, cb, pA3, pB3, n3, dist, impulse, life] (17 doubles); returns the point count */
int rpo_ref_manifolds(rpo_env* e, double* out, int max_points) {
  rpb_state* st = rpb_get(e);
  int n = 0;
  for (int i = 0; i < st->nman; i++)
    for (int j = 0; j < st->man[i].n && n < max_points; j++, n++) {
      const rpb_manifold* mf = &st->man[i]; const rpb_point* p = &mf->p[j];
      double* o = out + 17 * n;
      o[0] = mf->key_a; o[1] = mf->key_b; o[2] = mf->ca; o[3] = mf->cb;
      for (int k = 0; k < 3; k++) { o[4 + k] = p->pA[k]; o[7 + k] = p->pB[k]; o[10 + k] = p->n[k]; }
      o[13] = p->dist; o[14] = p->imp; o[15] = p->life; o[16] = mf->thr;
    }
  return n;
}
/* rows of the latest step touching arm dof `dof`: per row [index, J[dof], B[dof], rhs, dinv, lo, hi, lambda] */
int rpo_ref_rows_of_dof(rpo_env* e, int dof, double* out, int max_rows) {
  rpb_state* st = rpb_get(e);
  int n = 0;
  for (int i = 0; i < st->n_noncontact && n < max_rows; i++) {
    const rpb_row* r = &st->rows[i];
    if (r->J[dof] == 0) continue;
    double* o = out + 8 * n++;
    o[0] = i; o[1] = r->J[dof]; o[2] = r->B[dof]; o[3] = r->rhs; o[4] = r->dinv; o[5] = r->lo; o[6] = r->hi; o[7] = r->lambda;
  }
  return n;
}

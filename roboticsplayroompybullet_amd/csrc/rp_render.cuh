/* rp_render.cuh — headless camera images and batched ray queries over the collider tables.
 *
 * The reference renders obs['img'] with PyBullet's OpenGL rasteriser (environments.py:21-30, 841-845: getCameraImage(200, 200,
 * viewMatrix, projectionMatrix), fixed camera; gripper camera environments.py:33-49) and queries one ray per step for
 * gripper_proprioception (environments.py:720-743, rayTest).  Here both are ray casts against the SAME shapes the contact model uses - boxes, spheres and, for the arm's
 * links, the convex hulls of their collision meshes (round 5: until then their boxes) - the only geometry the library holds - with the colours of the reference's visual shapes (scenes.py rgbaColor, URDF
 * materials; the button's globe and the dial's grill recoloured as updateToggles does, environments.py:469-483), flat Lambert shading,
 * no textures, no shadows (the reference passes shadow=0).  Sub-goal visualisation (environments.py:606-690) = the same bodies drawn a
 * second time, half transparent, at the poses a sub-goal vector names.
 *
 *   k_collider_poses   one wave per env: FK -> world pose, half extents, colour of every collider (+ the ghosts of a sub-goal)
 *   k_render           256 pixels per block, the env's collider table staged in LDS, nearest hit per pixel
 *   k_ray_test         K rays per env against the same table: hit fraction, position, normal, collider, Bullet link index */
#pragma once

#define RC_STRIDE 20          /* floats per collider record: R9 p3 he3 rgb3 type|flags link */
#define RC_MAX (2 * RP_MAX_COL)
#define RC_GHOST 0x100        /* flag in the type word: drawn half transparent, ignored by rp_ray_test */
#define RC_HULL 0x200         /* flag in the type word: an arm link - the record's box is the box AROUND its collision hull, the shape is the hull (DevModel.hpl: its face planes) */

struct RpCamera { float eye[3], fwd[3], right[3], up[3]; float tan_half_fov, aspect; int mode; };   /* mode 1: at the EE link (gripper camera) */

/* world pose of every collider of env `first + blockIdx.x` into tab[blockIdx.x][RC_MAX][RC_STRIDE]; cnt[blockIdx.x] = records written.
 * sub_goal (may be null): [num][n_ag] achieved-goal vectors to visualise as ghosts (objects, drawer, door, button, dial).
 * ghost_arm (may be null): [num][8] = EE position, orientation quaternion, gripper: the ghost ARM of visualise_sub_goal's 'controllable_achieved_goal' /
 * 'full_positional_state' (environments.py:623-637, 671-674: reset_arm(ghost_arm, sub_goal, from_init=False) = the arm at its rest pose, one default IK call of 20
 * iterations towards the pose, joints [0:6] taken - reset_arm_goal_obs' arithmetic), drawn half transparent like the other ghosts. */
__global__ void __launch_bounds__(64) k_collider_poses(const DevModel* __restrict__ m, const float* __restrict__ state, int first, int num,
                                                       float* __restrict__ tab, int* __restrict__ cnt, const float* __restrict__ sub_goal,
                                                       float* __restrict__ ee_pose, const float* __restrict__ ghost_arm) {
  __shared__ EnvLds L;
  const int slot = blockIdx.x, lane = threadIdx.x;
  if (slot >= num) return;
  const int env = first + slot;
  load_state(L, state, env, lane);
  float* out = tab + (size_t)slot * RC_MAX * RC_STRIDE;
  const bool obj_ghosts = sub_goal && m->num_objects > 0;
  const int passes = (obj_ghosts || ghost_arm) ? 2 : 1;
  int nrec = 0;
  for (int pass = 0; pass < passes; pass++) {
    if (pass == 1 && ghost_arm) {      /* the ghost arm's joints: rest pose, one IK call, the first six joints (environments.py:575-590 with from_init's pose) */
      const float* ga = ghost_arm + 8 * (size_t)slot;
      __syncthreads();
      if (lane == 0) {
        const int nrest = m->arm_type == RP_ARM_PANDA ? 8 : 6;
        for (int i = 0; i < nrest; i++) L.st[ST_Q + i] = m->rest[i];
      }
      __syncthreads();
      ChainQ cur;
#pragma unroll
      for (int j = 0; j < 7; j++) cur.q[j] = j < m->ee_chain ? L.st[ST_Q + j] : 0.f;
      Q4 gq; gq.x = ga[3]; gq.y = ga[4]; gq.z = ga[5]; gq.w = ga[6];
      const ChainQ sol = ik_solve(m, mk3(ga[0], ga[1], ga[2]), gq, cur, 20, lane & 15);
      __syncthreads();
      if (lane == 0) for (int i = 0; i < 6; i++) L.st[ST_Q + i] = sol.q[i];
      __syncthreads();
      if (ee_pose && lane < 8) ee_pose[12 * (size_t)num + 8 * (size_t)slot + lane] = L.st[ST_Q + lane];      /* (test hook rp_debug_ghost_joints: the ghost's joints behind the EE poses) */
    }
    if (pass == 1 && obj_ghosts) {           /* the sub-goal's poses: objects [pos3 (quat4)] ..., then drawer y, door, button, dial (play ids) */
      __syncthreads();
      if (lane == 0) {
        const float* g = sub_goal + (size_t)slot * m->n_ag;
        int idx = 0;
        for (int b = 0; b < m->num_objects; b++) {
          float* f = &L.st[ST_FREE + 13 * b];
          for (int k = 0; k < 3; k++) f[k] = g[idx + k];
          idx += 3;
          if (m->use_orientation) { for (int k = 0; k < 4; k++) f[3 + k] = g[idx + k]; idx += 4; }
        }
        if (m->play) {
          L.st[ST_FREE + 13 * m->drawer_free + 1] = g[idx];
          for (int k = 0; k < m->n_j1; k++) L.st[ST_JQ + k] = g[idx + 1 + k];
        }
      }
      __syncthreads();
    }
    fk_bodies(m, L, lane);
    __syncthreads();
    if (pass == 0 && ee_pose && lane == 0) {        /* EE link pose for the gripper camera */
      const int eb = m->site_body[RP_SITE_EE];
      const M3 Rb = ldm3(&L.xR[9 * eb]);
      const V3 pos = ld3(&L.xp[3 * eb]) + mulv(Rb, ld3(m->site_pos[RP_SITE_EE]));
      const M3 Rs = mul(Rb, ldm3(m->site_rot[RP_SITE_EE]));
      float* e = ee_pose + 12 * slot;
      st3(e, pos);
      for (int k = 0; k < 9; k++) e[3 + k] = Rs.m[k];
    }
    const int cbody = lane < m->n_col ? m->col_body[lane] : 0;
    const bool mine = lane < m->n_col && (pass == 0 || (obj_ghosts && cbody > m->n_arm) || (ghost_arm && cbody >= 1 && cbody <= m->n_arm));    /* ghosts: free bodies and scene joints of a sub-goal, the arm's links of a ghost arm */
    const unsigned long long bal = __ballot(mine);
    if (mine) {
      const int r = nrec + __popcll(bal & ((1ull << lane) - 1ull));
      const Xf x = collider_xf(m, L, lane);
      float* o = out + r * RC_STRIDE;
      for (int k = 0; k < 9; k++) o[k] = x.R.m[k];
      st3(o + 9, x.p);
      st3(o + 12, ld3(m->col_he[lane]));
      float cr = m->col_rgb[lane][0], cg = m->col_rgb[lane][1], cb = m->col_rgb[lane][2];
      const int tog = m->col_toggle[lane];
      if (tog == 1 && m->n_j1 > 1) { const bool on = L.st[ST_JQ + 1] < 0.025f; cr = 1.f; cg = on ? 0.f : 1.f; cb = on ? 0.f : 1.f; }
      if (tog == 2 && m->n_j1 > 2) { const bool on = dial01(L.st[ST_JQ + 2]) < 0.5f; cr = 1.f; cg = on ? 0.f : 1.f; cb = on ? 0.f : 1.f; }
      o[15] = cr; o[16] = cg; o[17] = cb;
      o[18] = __int_as_float(m->col_type[lane] | (pass == 1 ? RC_GHOST : 0) | ((m->hpl && m->hpl_cnt[lane] > 0) ? RC_HULL : 0));
      o[19] = __int_as_float((m->col_link[lane] << 8) | lane);
    }
    nrec += __popcll(bal);
  }
  if (lane == 0) cnt[slot] = nrec;
}

/* ray o + t d, t in [0, tmax], against a box (R, p, he): entry parameter and the face normal, world frame.  A ray that starts
 * inside reports no hit (Bullet's convex ray test). */
__device__ __forceinline__ bool rc_ray_box(const float* rec, V3 o, V3 d, float tmax, float& t_hit, V3& n_hit) {
  const M3 R = ldm3(rec);
  const V3 ol = tmulv(R, o - ld3(rec + 9)), dl = tmulv(R, d);
  const float o3[3] = {ol.x, ol.y, ol.z}, d3[3] = {dl.x, dl.y, dl.z}, h3[3] = {rec[12], rec[13], rec[14]};
  if (fabsf(o3[0]) <= h3[0] && fabsf(o3[1]) <= h3[1] && fabsf(o3[2]) <= h3[2]) return false;
  float tmin = 0.f, tm = tmax; int axis = 0; float sgn = 0.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (fabsf(d3[k]) < 1e-12f) { if (fabsf(o3[k]) > h3[k]) return false; continue; }
    float t1 = (-h3[k] - o3[k]) / d3[k], t2 = (h3[k] - o3[k]) / d3[k], s = -1.f;
    if (t1 > t2) { float w = t1; t1 = t2; t2 = w; s = 1.f; }
    if (t1 > tmin) { tmin = t1; axis = k; sgn = s; }
    tm = fminf(tm, t2);
    if (tmin > tm) return false;
  }
  t_hit = tmin;
  n_hit = col(R, axis) * sgn;
  return true;
}
/* ... against an arm link: the convex hull of its collision mesh - what the link collides as (rp_kernels.cuh hull_item16) - by clipping the ray against the hull's face
 * planes (collider frame, n . x + w <= 0 inside), after the box around the hull, which most rays miss.  The reference draws the links' visual meshes; rounds 3 and 4 drew the
 * boxes.  A ray that starts inside reports no hit, like the box's. */
__device__ __forceinline__ bool rc_ray_hull(const float* rec, const float4* __restrict__ pl, int npl, V3 o, V3 d, float tmax, float& t_hit, V3& n_hit) {
  const M3 R = ldm3(rec);
  const V3 ol = tmulv(R, o - ld3(rec + 9)), dl = tmulv(R, d);
  {      /* the box around the hull as a SLAB-OVERLAP cull only: a ray that starts inside the box but outside the hull (the gripper camera at the EE link, a ray cast from
          * beside the arm: the links' boxes are much larger than their hulls) still meets the planes below - rc_ray_box's "starts inside = no hit" is about the hull */
    const float o3[3] = {ol.x, ol.y, ol.z}, d3[3] = {dl.x, dl.y, dl.z}, h3[3] = {rec[12], rec[13], rec[14]};
    float tmin = 0.f, tm = tmax;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (fabsf(d3[k]) < 1e-12f) { if (fabsf(o3[k]) > h3[k]) return false; continue; }
      float t1 = (-h3[k] - o3[k]) / d3[k], t2 = (h3[k] - o3[k]) / d3[k];
      if (t1 > t2) { const float w = t1; t1 = t2; t2 = w; }
      tmin = fmaxf(tmin, t1); tm = fminf(tm, t2);
      if (tmin > tm) return false;
    }
  }
  float t_in = 0.f, t_out = tmax; V3 nl = mk3(0, 0, 0); bool entered = false;
  for (int k = 0; k < npl; k++) {
    const float4 p = pl[k];
    const float den = p.x * dl.x + p.y * dl.y + p.z * dl.z, num = -(p.x * ol.x + p.y * ol.y + p.z * ol.z + p.w);
    if (fabsf(den) < 1e-12f) { if (num < 0.f) return false; continue; }      /* parallel to the plane: outside it = a miss */
    const float t = num / den;
    if (den < 0.f) { if (t > t_in) { t_in = t; nl = mk3(p.x, p.y, p.z); entered = true; } }
    else t_out = fminf(t_out, t);
    if (t_in > t_out) return false;
  }
  if (!entered) return false;
  t_hit = t_in;
  n_hit = mulv(R, nl);
  return true;
}
__device__ __forceinline__ bool rc_ray_sphere(const float* rec, V3 o, V3 d, float tmax, float& t_hit, V3& n_hit) {
  const V3 c = ld3(rec + 9), oc = o - c;
  const float r = rec[12], a = dot(d, d), b = 2.f * dot(oc, d), cc = dot(oc, oc) - r * r;
  if (cc < 0.f) return false;
  const float disc = b * b - 4.f * a * cc;
  if (disc < 0.f) return false;
  const float t = (-b - sqrtf(disc)) / (2.f * a);
  if (t < 0.f || t > tmax) return false;
  t_hit = t;
  n_hit = (oc + d * t) * (1.f / r);
  return true;
}

/* obs['img'] (environments.py:841-845): one thread per pixel, row 0 = top of the image, uint8 rgb */
__global__ void __launch_bounds__(256) k_render(const DevModel* __restrict__ m, const float* __restrict__ tab, const int* __restrict__ cnt, int num, RpCamera cam, const float* __restrict__ ee_pose,
                                                int width, int height, unsigned char* __restrict__ rgb) {
  __shared__ float T[RC_MAX * RC_STRIDE];
  const int tiles = (width * height + 255) / 256;
  const int slot = blockIdx.x / tiles, tile = blockIdx.x - slot * tiles;
  if (slot >= num) return;
  const int n = cnt[slot];
  for (int i = threadIdx.x; i < n * RC_STRIDE; i += 256) T[i] = tab[(size_t)slot * RC_MAX * RC_STRIDE + i];
  __syncthreads();
  const int pix = tile * 256 + threadIdx.x;
  if (pix >= width * height) return;
  const int px = pix % width, py = pix / width;
  V3 eye = ld3(cam.eye), fwd = ld3(cam.fwd), right = ld3(cam.right), up = ld3(cam.up);
  if (cam.mode == 1) {       /* gripper_camera (environments.py:33-49): at the EE link, looking along its x axis pitched by -90 deg, up = its z */
    const float* e = ee_pose + 12 * slot;
    const M3 R = ldm3(e + 3);
    eye = ld3(e);
    /* ori = euler(link) + (0, -pi/2, 0) in the reference; restated as the link frame's axes: forward = -z, up = x */
    fwd = col(R, 2) * -1.f; up = col(R, 0); right = cross(fwd, up);
  }
  const float nx = (2.f * (px + 0.5f) / width - 1.f) * cam.tan_half_fov * cam.aspect, ny = (1.f - 2.f * (py + 0.5f) / height) * cam.tan_half_fov;
  V3 d = fwd + right * nx + up * ny;
  d = d * (1.f / norm(d));
  const float far = 10.f;
  float best = far, gbest = far; V3 bn = mk3(0, 0, 1), gn = bn; int bi = -1, gi = -1;
  for (int c = 0; c < n; c++) {
    const float* rec = &T[c * RC_STRIDE];
    const int tw = __float_as_int(rec[18]);
    float t; V3 nn;
    const int ci = __float_as_int(rec[19]) & 0xFF;
    const bool hit = (tw & RC_HULL) ? rc_ray_hull(rec, (const float4*)m->hpl + m->hpl_off[ci], m->hpl_cnt[ci], eye, d, far, t, nn)
                                    : ((tw & 0xFF) == 0 ? rc_ray_box(rec, eye, d, far, t, nn) : rc_ray_sphere(rec, eye, d, far, t, nn));
    if (!hit) continue;
    if (tw & RC_GHOST) { if (t < gbest) { gbest = t; gn = nn; gi = c; } }
    else if (t < best) { best = t; bn = nn; bi = c; }
  }
  const V3 light = mk3(0.2672612f, -0.5345225f, 0.8017837f);          /* (1, -2, 3) / sqrt(14): fixed directional light */
  auto shade = [&](int c, V3 nn) {
    const float* rec = &T[c * RC_STRIDE];
    const float k = 0.45f + 0.55f * fmaxf(0.f, dot(nn, light));
    return mk3(rec[15] * k, rec[16] * k, rec[17] * k);
  };
  V3 colr = mk3(0.82f, 0.88f, 0.96f);                                    /* background */
  if (bi >= 0) colr = shade(bi, bn);
  if (gi >= 0 && gbest < best) colr = colr * 0.5f + shade(gi, gn) * 0.5f;  /* the ghosts are drawn with alpha 0.5 (environments.py:650) */
  unsigned char* o = rgb + ((size_t)slot * width * height + pix) * 3;
  o[0] = (unsigned char)(fminf(fmaxf(colr.x, 0.f), 1.f) * 255.f + 0.5f);
  o[1] = (unsigned char)(fminf(fmaxf(colr.y, 0.f), 1.f) * 255.f + 0.5f);
  o[2] = (unsigned char)(fminf(fmaxf(colr.z, 0.f), 1.f) * 255.f + 0.5f);
}

/* rayTest (environments.py:738-741) for K rays per env: from / to [num][K][3]; fraction 1 and collider -1 on a miss */
__global__ void __launch_bounds__(64) k_ray_test(const DevModel* __restrict__ m, const float* __restrict__ tab, const int* __restrict__ cnt, int num, int K, const float* __restrict__ from,
                                                 const float* __restrict__ to, float* __restrict__ frac, int* __restrict__ collider, int* __restrict__ link,
                                                 float* __restrict__ hit_pos, float* __restrict__ hit_nrm) {
  __shared__ float T[RC_MAX * RC_STRIDE];
  const int slot = blockIdx.x;
  if (slot >= num) return;
  const int n = cnt[slot];
  for (int i = threadIdx.x; i < n * RC_STRIDE; i += 64) T[i] = tab[(size_t)slot * RC_MAX * RC_STRIDE + i];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 64) {
    const size_t r = (size_t)slot * K + k;
    const V3 o = ld3(from + 3 * r), d = ld3(to + 3 * r) - o;
    float best = 2.f; V3 bn = mk3(0, 0, 0); int bi = -1;
    for (int c = 0; c < n; c++) {
      const float* rec = &T[c * RC_STRIDE];
      const int tw = __float_as_int(rec[18]);
      if (tw & RC_GHOST) continue;
      float t; V3 nn;
      const int ci = __float_as_int(rec[19]) & 0xFF;
      const bool hit = (tw & RC_HULL) ? rc_ray_hull(rec, (const float4*)m->hpl + m->hpl_off[ci], m->hpl_cnt[ci], o, d, 1.f, t, nn)
                                      : ((tw & 0xFF) == 0 ? rc_ray_box(rec, o, d, 1.f, t, nn) : rc_ray_sphere(rec, o, d, 1.f, t, nn));
      if (hit && t < best) { best = t; bn = nn; bi = c; }            /* the lowest collider index wins ties, as in calc_state's ray */
    }
    const bool hit = bi >= 0 && best <= 1.f;
    if (frac) frac[r] = hit ? best : 1.f;
    const int id = hit ? __float_as_int(T[bi * RC_STRIDE + 19]) : -1;
    if (collider) collider[r] = hit ? (id & 0xFF) : -1;
    if (link) link[r] = hit ? (id >> 8) : -1;
    if (hit_pos) st3(hit_pos + 3 * r, hit ? o + d * best : ld3(to + 3 * r));
    if (hit_nrm) st3(hit_nrm + 3 * r, hit ? bn : mk3(0, 0, 0));
  }
}

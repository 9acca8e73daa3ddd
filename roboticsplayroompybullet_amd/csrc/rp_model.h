/* rp_model.h — baked model tables for one env kind = arm + scene (U = UR5PlayAbsRPY1Obj-v0, R = UR5Reach-v0,
 * P = pandaPick-v0, Q = pandaReach-v0, V = pandaPlayAbsRPY1Obj-v0).  Plain C, shared as DATA LAYOUT by oracle/rp_oracle.c (CPU restatement) and the
 * HIP library; filled by generated/rp_models_gen.h (tools/bake_assets.py).
 *
 * Bodies of one env:   0 = static world (scene statics + the arm's fixed base link)
 *                      1 .. n_arm                 movable arm links (fixed URDF links merged in), dof i-1
 *                      n_arm+1 .. n_arm+n_free    free 6-DoF bodies  (U, V: block, drawer; W: block, block, drawer; P: block)
 *                      then n_joint1 single-DoF scene joints on a static base (U: door, button, dial)
 * Generalised velocity: [arm n_arm | free 6 each (lin xyz, ang xyz, world frame) | joint1 1 each].
 * Reference anchors: arm tables <- ur5e2.urdf / panda.urdf; scene <- scenes.py:46-426 as
 * captured in tests/golden/scenes.json (App. C). */
#ifndef RP_MODEL_H
#define RP_MODEL_H

#define RP_KIND_U 0
#define RP_KIND_R 1
#define RP_KIND_P 2
#define RP_KIND_Q 3   /* Panda + default_scene (pandaReach-v0, pandaReach2D-v0) */
#define RP_KIND_V 4   /* Panda + complex_scene, one block (the pandaPlay*1Obj-v0 ids) */
#define RP_KIND_W 5   /* Panda + complex_scene, two blocks (pandaPlay-v0, pandaPlayJoints-v0) */
#define RP_N_KIND 6

#define RP_ARM_UR5 0
#define RP_ARM_PANDA 1
#define RP_SCENE_COMPLEX 0
#define RP_SCENE_DEFAULT 1
#define RP_SCENE_PUSH 2

#define RP_MAX_ARM 12
#define RP_MAX_FREE 3
#define RP_MAX_J1 3
#define RP_MAX_COL 64
#define RP_MAX_PAIR 1024
#define RP_MAX_SITE 4
#define RP_MAX_NV (RP_MAX_ARM + 6 * RP_MAX_FREE + RP_MAX_J1)

#define RP_SITE_EE 0     /* Bullet endEffectorIndex (UR5 link 7 / Panda link 11), COM frame */
#define RP_SITE_WRIST 1  /* endEffectorIndex-1 (UR5 only; gripper_proprioception, environments.py:725) */
#define RP_SITE_PADL 2   /* UR5 link 18 */
#define RP_SITE_PADR 3   /* UR5 link 20 */

typedef struct rp_model {
  int kind, n_arm, n_free, n_joint1, n_col, n_pair, n_site;
  int arm_type, scene;              /* RP_ARM_*, RP_SCENE_*: what the kind is made of */
  int drawer_free;                  /* index of the drawer among the free bodies (the objects come first), -1 = none */
  int free_row0;                    /* bit f: free body f shares the arm's half of the solver's velocity layout (W: the drawer); contacts
                                     * are ordered by which halves they touch (see collide / rp_oracle.c) */
  /* arm (tree, parents precede children) */
  int arm_parent[RP_MAX_ARM];       /* movable parent (0-based) or -1 = base */
  int arm_jtype[RP_MAX_ARM];        /* 0 revolute, 1 prismatic */
  int arm_bullet_index[RP_MAX_ARM]; /* Bullet joint index of this dof */
  double arm_jpos[RP_MAX_ARM][3];   /* joint frame origin in parent body frame */
  double arm_jrot[RP_MAX_ARM][9];   /* joint frame rotation, row-major, v_parent = R v_child (at q = 0) */
  double arm_axis[RP_MAX_ARM][3];   /* unit joint axis in child frame */
  double arm_mass[RP_MAX_ARM];
  double arm_com[RP_MAX_ARM][3];    /* COM in body frame */
  double arm_inertia[RP_MAX_ARM][9];/* about COM, body frame */
  double arm_lower[RP_MAX_ARM], arm_upper[RP_MAX_ARM];
  double base_pos[3], base_rot[9];
  double rest[RP_MAX_ARM];
  int site_body[RP_MAX_SITE];
  double site_pos[RP_MAX_SITE][3], site_rot[RP_MAX_SITE][9];
  /* free bodies */
  double free_mass[RP_MAX_FREE], free_inertia[RP_MAX_FREE][3], free_pos0[RP_MAX_FREE][3], free_rot0[RP_MAX_FREE][9];
  int free_rot_locked[RP_MAX_FREE]; /* zero inertia => no angular response (drawer) */
  /* single-dof scene joints */
  int j1_type[RP_MAX_J1];
  double j1_pos[RP_MAX_J1][3], j1_rot[RP_MAX_J1][9], j1_axis[RP_MAX_J1][3];
  double j1_mass[RP_MAX_J1], j1_inertia_axis[RP_MAX_J1];
  int j1_has_pos_motor[RP_MAX_J1];
  double j1_motor_target[RP_MAX_J1], j1_motor_force[RP_MAX_J1];
  /* colliders (type 0 box, 1 sphere [he[0] = radius]); pose in the owning body's frame (world for body 0) */
  int col_body[RP_MAX_COL], col_type[RP_MAX_COL];
  int col_link[RP_MAX_COL];         /* Bullet link index of an arm collider (-1 otherwise); rayTest link filter */
  int col_obj[RP_MAX_COL];          /* collision object id: one contact manifold (<= 4 points) per object pair */
  double col_he[RP_MAX_COL][3], col_pos[RP_MAX_COL][3], col_rot[RP_MAX_COL][9], col_friction[RP_MAX_COL];
  /* Bullet's contact breaking threshold of the collision object the collider belongs to: gContactBreakingThreshold (0.02) x
   * btCollisionShape::getAngularMotionDisc (|AABB centre| + half diagonal of the object's shape in its own frame); a pair's
   * threshold is the smaller of its two objects' (btCollisionDispatcher::getNewManifold, relative thresholds are its default) */
  double col_thr[RP_MAX_COL];
  /* URDF <contact> stiffness / damping of the link the collider belongs to (0 = absent): Bullet turns them into the contact row's
   * cfm and erp (BT_CONTACT_FLAG_CONTACT_STIFFNESS_DAMPING, btMultiBodyConstraintSolver::setupMultiBodyContactConstraint) */
  double col_stiffness[RP_MAX_COL], col_damping[RP_MAX_COL];
  double col_spin[RP_MAX_COL];      /* URDF <contact> spinning_friction of the link (0 = absent): torsional friction of its contacts */
  /* rendering (environments.py:841-845 img): colour of the visual shape the collider stands for (scenes.py rgbaColor, URDF materials);
   * col_toggle 1 = the globe recoloured by the button, 2 = the grill recoloured by the dial (updateToggles, environments.py:469-483) */
  double col_rgb[RP_MAX_COL][3];
  int col_toggle[RP_MAX_COL];
  /* candidate collider pairs (first = collider of the higher body id), sorted so that the pairs of one
   * object pair (manifold) are contiguous */
  unsigned char pair[RP_MAX_PAIR][2];
} rp_model;

#endif

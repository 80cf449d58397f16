/*
 * rl_env.h - C-ABI of the MI355X-native vectorised locomotion RL environment.
 *
 * The reference (fan-ziqi/robot_lab v2.3.2) has NO FFI: its hot path is reached through
 * `gym.make(id, cfg=env_cfg)` -> `isaaclab.envs:ManagerBasedRLEnv(cfg)`
 * (source/robot_lab/robot_lab/tasks/manager_based/locomotion/velocity/config/quadruped/unitree_a1/__init__.py:12-32,
 *  scripts/reinforcement_learning/rsl_rl/train.py:177) and `env.step(action)` / `env.reset()`
 * (scripts/tools/zero_agent.py:56-73).  This header is the boundary a native binding of that
 * class would use (last row): plain pointers and sizes, no torch types.
 * The Python class `robot_lab_amd.env.ManagerBasedRLEnv` is the binding; INTEGRATION.md shows it.
 *
 * All device buffers are env-owned, device-resident, valid until rl_env_destroy, and are
 * rewritten in place by every rl_env_step - EXCEPT the two observation groups, which alternate between
 * two HBM buffers: the observations returned by step t stay untouched until step t + 2 is launched
 * (the reference builds fresh observation tensors on every step - ObservationManager.compute ->
 * torch.cat - and rsl_rl's PPO.act keeps a reference to them across the following env.step;
 * SURVEY.md 8(b) "Ownership").  All calls are stream-ordered on the hipStream_t passed in; no call
 * synchronises the host except rl_env_read_log and rl_env_import_state.
 *
 * Return value: 0 on success, negative on error; rl_env_last_error() gives the message.
 */
#ifndef RL_ENV_H_
#define RL_ENV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RL_MAX_LINKS 33
#define RL_MAX_DOF 32
#define RL_MAX_BODIES 48
#define RL_MAX_SPHERES 96
#define RL_MAX_CAPSULES 16   /* self-collision proxies: at most one capsule per selected link */
#define RL_MAX_SELF_PAIRS 80 /* capsule pairs tested per substep (5 per lane of the 16-lane env) */
#define RL_MAX_REWARD_TERMS 40
#define RL_MAX_OBS_TERMS 12
#define RL_TERM_NPARAM 8

/* ---- reward term kinds (each cites the reference function it restates) ------------------- */
enum rl_reward_kind {
  RL_REW_TRACK_LIN_VEL_XY_EXP = 0,  /* VEL/mdp/rewards.py:22-35   p0 = std^2 */
  RL_REW_TRACK_ANG_VEL_Z_EXP = 1,   /* rewards.py:38-48           p0 = std^2 */
  RL_REW_LIN_VEL_Z_L2 = 2,          /* rewards.py:647-653 */
  RL_REW_ANG_VEL_XY_L2 = 3,         /* rewards.py:656-662 */
  RL_REW_JOINT_TORQUES_L2 = 4,      /* [UPSTREAM isaaclab.envs.mdp] velocity_env_cfg.py:401-403 ; joint mask */
  RL_REW_JOINT_ACC_L2 = 5,          /* [UPSTREAM] velocity_env_cfg.py:407-409 ; joint mask */
  RL_REW_JOINT_POS_LIMITS = 6,      /* [UPSTREAM] velocity_env_cfg.py:419-421 ; joint mask */
  RL_REW_JOINT_POWER = 7,           /* rewards.py:81-90 ; joint mask */
  RL_REW_STAND_STILL = 8,           /* rewards.py:93-104 ; p0 = command_threshold ; joint mask */
  RL_REW_JOINT_POS_PENALTY = 9,     /* rewards.py:107-129 ; p0 stand_still_scale p1 velocity_threshold p2 command_threshold */
  RL_REW_JOINT_MIRROR = 10,         /* rewards.py:259-278 ; pairs in idx_a/idx_b, p0 = 1/len(mirror_joints) */
  RL_REW_ACTION_RATE_L2 = 11,       /* [UPSTREAM] velocity_env_cfg.py:506 */
  RL_REW_UNDESIRED_CONTACTS = 12,   /* rewards.py:665-675 ; p0 threshold ; body mask */
  RL_REW_CONTACT_FORCES = 13,       /* [UPSTREAM] velocity_env_cfg.py:519-523 ; p0 threshold ; body mask */
  RL_REW_FEET_CONTACT_WITHOUT_CMD = 14, /* rewards.py:416-425 ; body mask */
  RL_REW_FEET_HEIGHT_BODY = 15,     /* rewards.py:527-554 ; p0 target_height p1 tanh_mult ; body mask */
  RL_REW_UPWARD = 16,               /* rewards.py:608-613 */
  RL_REW_FEET_AIR_TIME = 17,        /* rewards.py:340-360 ; p0 threshold ; body mask */
  RL_REW_FEET_AIR_TIME_VARIANCE = 18, /* rewards.py:386-397 ; body mask */
  RL_REW_FEET_SLIDE = 19,           /* rewards.py:557-587 ; body mask */
  RL_REW_FEET_GAIT = 20,            /* rewards.py:156-256 ; p0 std p1 max_err p2 velocity_threshold p3 command_threshold ; idx_a = (pair0a,pair0b,pair1a,pair1b) */
  RL_REW_FLAT_ORIENTATION_L2 = 21,  /* rewards.py:678-687 */
  RL_REW_IS_TERMINATED = 22,        /* [UPSTREAM] velocity_env_cfg.py:379 */
  RL_REW_JOINT_DEVIATION_L1 = 23,   /* [UPSTREAM] velocity_env_cfg.py:411-417 ; joint mask */
  RL_REW_JOINT_VEL_L2 = 24,         /* [UPSTREAM] velocity_env_cfg.py:404-406 ; joint mask */
  RL_REW_FEET_CONTACT = 25,         /* rewards.py:399-413 ; p0 expect_contact_num ; body mask */
  RL_REW_FEET_STUMBLE = 26,         /* rewards.py:428-436 ; body mask */
  RL_REW_FEET_HEIGHT = 27,          /* rewards.py:507-524 ; p0 target_height p1 tanh_mult ; body mask */
  RL_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP = 28, /* rewards.py:51-66 ; p0 = std^2 (G1) */
  RL_REW_TRACK_ANG_VEL_Z_WORLD_EXP = 29,      /* rewards.py:69-78 ; p0 = std^2 (G1) */
  RL_REW_FEET_AIR_TIME_POSITIVE_BIPED = 30,   /* rewards.py:363-383 ; p0 threshold ; body mask (G1) */
  RL_REW_HANDSTAND_FEET_HEIGHT_EXP = 31,      /* config/others/unitree_a1_handstand/env/rewards.py:18-28 ; p0 std^2 p1 target_height ; body mask */
  RL_REW_HANDSTAND_FEET_ON_AIR = 32,          /* .../env/rewards.py:31-37 ; body mask */
  RL_REW_HANDSTAND_FEET_AIR_TIME = 33,        /* .../env/rewards.py:40-47 ; p0 threshold ; body mask */
  RL_REW_HANDSTAND_ORIENTATION_L2 = 34,       /* .../env/rewards.py:50-59 ; p0..p2 target gravity */
  RL_REW_BASE_HEIGHT_L2 = 35,                 /* rewards.py:616-644 ; p0 target_height p1 = 1: target follows the mean hit height of the
                                                 3 x 3 base ray caster (velocity_env_cfg.py:78-85), 0: world height */
  RL_REW_WHEEL_VEL_PENALTY = 36,              /* rewards.py:132-153 ; p0 velocity_threshold p1 command_threshold ;
                                                 pairs (idx_a = wheel body, idx_b = wheel joint), n_idx of them */
  RL_REW_FEET_DISTANCE_Y_EXP = 37,            /* rewards.py:439-461 ; p0 std^2 p1 stance_width ; idx_a = feet in asset_cfg order */
  RL_REW_FEET_DISTANCE_XY_EXP = 38,           /* rewards.py:464-505 ; p0 std^2 p1 stance_width p2 stance_length ; idx_a = the 4 feet */
  RL_REW_ACTION_MIRROR = 39,                  /* rewards.py:281-302 ; pairs of ACTION columns in idx_a/idx_b (the reference indexes the action vector
                                                 with the joints' indices), p0 = 1/len(mirror_joints): sum (|a_i| - |a_j|)^2 */
  RL_REW_ACTION_SYNC = 40,                    /* rewards.py:305-337 ; idx_a = the action columns of all joint groups one after the other, idx_b = the group
                                                 of each entry (groups of fewer than two joints are not listed), p0 = 1/len(joint_groups):
                                                 sum over the groups of the (biased) variance of |a| inside the group */
  RL_REW_NUM_KINDS
};

/* ---- observation term kinds (velocity_env_cfg.py:134-254) -------------------------------- */
enum rl_obs_kind {
  RL_OBS_BASE_LIN_VEL = 0,
  RL_OBS_BASE_ANG_VEL = 1,
  RL_OBS_PROJECTED_GRAVITY = 2,
  RL_OBS_VELOCITY_COMMANDS = 3,
  RL_OBS_JOINT_POS_REL = 4,
  RL_OBS_JOINT_VEL_REL = 5,
  RL_OBS_LAST_ACTION = 6,
  RL_OBS_HEIGHT_SCAN = 7,
  RL_OBS_JOINT_POS_REL_NO_WHEEL = 8 /* VEL/mdp/observations.py:17-27 */
};

typedef struct rl_reward_term {
  int32_t kind;
  float weight;
  float p[RL_TERM_NPARAM];
  uint32_t joint_mask;   /* bit j: joint j (task joint order) takes part */
  uint64_t body_mask;    /* bit b: body b takes part */
  int32_t idx_a[16];     /* kind-specific index lists (mirror pairs, gait feet) */
  int32_t idx_b[16];
  int32_t n_idx;
} rl_reward_term;

typedef struct rl_obs_term {
  int32_t kind;
  float scale;
  float clip_lo, clip_hi;
  float noise_lo, noise_hi; /* additive uniform noise; applied only if the group corrupts */
  int32_t has_noise;
} rl_obs_term;

/* ---- articulated model: what the reference passes as ArticulationCfg + URDF -------------- */
typedef struct rl_model_desc {
  int32_t num_links;    /* moving rigid bodies incl. base (A1: 13) */
  int32_t num_dof;      /* actuated joints, task joint order (A1: 12) */
  int32_t num_bodies;   /* sensor/randomisation bodies (A1: 17) */
  int32_t num_spheres;  /* collision spheres */
  /* topology the lane program simulates: a serial "trunk" chain of num_trunk joints hanging off the
     base (G1: the 3 waist joints pelvis -> torso; quadrupeds: none) and num_chains = 4 serial limb
     chains, chain k hanging off the base (chain_attach = 0) or off the trunk link reached after
     chain_attach[k] trunk joints (G1 arms: 3 = torso).  0 chains = topology not supported. */
  int32_t num_chains;
  int32_t chain_len;                    /* longest chain (A1 3, Go2W 4, G1 7) */
  int32_t chain_link[4][8];             /* link index of chain k's j-th link (joint index = link - 1), -1 padded */
  int32_t chain_nj[4];                  /* joints of chain k (G1: legs 6, arms 7) */
  int32_t chain_attach[4];
  int32_t num_trunk;
  int32_t trunk_link[8];                /* <= 6 in use (GR1: waist 3 + head 3, the arms leaving the spine at depth 3); see trunk_parent */
  int32_t link_parent[RL_MAX_LINKS];
  float link_origin[RL_MAX_LINKS][3];   /* joint origin in parent link frame */
  float link_quat[RL_MAX_LINKS][4];     /* joint frame orientation in the parent link frame, (w,x,y,z) (URDF joint rpy) */
  float link_axis[RL_MAX_LINKS][3];
  float joint_lower[RL_MAX_DOF], joint_upper[RL_MAX_DOF];
  float joint_vel_limit[RL_MAX_DOF];
  float joint_armature[RL_MAX_DOF];
  float default_joint_pos[RL_MAX_DOF], default_joint_vel[RL_MAX_DOF];
  float soft_lower[RL_MAX_DOF], soft_upper[RL_MAX_DOF];
  int32_t body_link[RL_MAX_BODIES];
  float body_pos[RL_MAX_BODIES][3];      /* body frame origin in its link frame */
  float body_mass[RL_MAX_BODIES];
  float body_com[RL_MAX_BODIES][3];      /* link frame */
  float body_inertia[RL_MAX_BODIES][6];  /* about com, link axes: xx yy zz xy xz yz */
  int32_t sphere_body[RL_MAX_SPHERES];
  float sphere_center[RL_MAX_SPHERES][3]; /* link frame */
  float sphere_radius[RL_MAX_SPHERES];
  float default_root_pos[3];
  float default_root_quat[4];            /* w x y z */
  /* actuators (unitree.py:55-63 DCMotorCfg ; :158-173 ImplicitActuatorCfg) */
  int32_t act_implicit[RL_MAX_DOF];      /* 0 = explicit DC motor, 1 = implicit PD */
  float act_kp[RL_MAX_DOF], act_kd[RL_MAX_DOF];
  float act_effort_limit[RL_MAX_DOF], act_saturation[RL_MAX_DOF], act_vel_limit[RL_MAX_DOF];
  /* action terms (velocity_env_cfg.py:124-126 ; unitree_go2w/rough_env_cfg.py:29-31) */
  int32_t action_is_vel[RL_MAX_DOF];
  float action_scale[RL_MAX_DOF], action_offset[RL_MAX_DOF];
  float action_clip_lo[RL_MAX_DOF], action_clip_hi[RL_MAX_DOF];
  /* self-collision (ArticulationRootPropertiesCfg.enabled_self_collisions: unitree.py:482 G1, roboparty.py:33 ATOM01; off
     everywhere else).  The reference hands the links' collision meshes to PhysX, which collides every pair of links except
     parent / child; here each of up to RL_MAX_CAPSULES links carries ONE capsule (segment p0 - p1 in the link frame + radius,
     fitted to the link's collision geometry) and the listed capsule pairs repel each other with an explicit penalty force
     (rl_sim_desc.self_k) once per substep.  0 capsules / 0 pairs = no self-collision (trunk + limbs instance only). */
  int32_t self_collision;                /* the cfg's flag, as read */
  int32_t num_capsules;
  int32_t capsule_link[RL_MAX_CAPSULES];
  float capsule_p0[RL_MAX_CAPSULES][3], capsule_p1[RL_MAX_CAPSULES][3];
  float capsule_radius[RL_MAX_CAPSULES];
  int32_t num_self_pairs;
  int32_t self_pair[RL_MAX_SELF_PAIRS][2]; /* capsule indices a < b */
  /* Collision spheres of a trunk link are evaluated by the lanes whose "link group 0" rides on that link - by default the lanes of the
     limbs that hang off it (chain_attach).  A spine link nothing hangs off (FFTAI GR1: the head, behind the torso the arms leave from)
     gets the group 0 of a lane that shares its own attachment link with another lane: chain_grp0[k] = 1 + trunk depth of the link
     chain k's group 0 rides on, 0 = the default (the chain's attachment link). */
  int32_t chain_grp0[4];
  /* The trunk is a serial spine (G1: the waist; GR1: waist + neck) or several serial PIECES that each start at the base (Booster T1:
     the waist, which carries the legs, and the two-joint neck both hang off the trunk body).  trunk_parent[i]: 0 = trunk joint i hangs
     off its predecessor in trunk_link[] (joint 0: off the base) - the serial spine, and what a zeroed field means; -1 = it hangs off
     the base and starts a new piece.  Nothing else. */
  int32_t trunk_parent[8];
} rl_model_desc;

/* ---- simulator constants (ours; the reference delegates these to PhysX) ------------------ */
typedef struct rl_sim_desc {
  float dt;               /* velocity_env_cfg.py:717 */
  int32_t decimation;     /* :714 */
  float gravity;          /* 9.81 */
  float contact_k;        /* normal stiffness N/m */
  float contact_c;        /* normal damping N s/m at full engagement */
  float contact_phi_ref;  /* penetration at which damping is fully engaged */
  float contact_ct;       /* tangential (stick) damping N s/m */
  float contact_vdep;     /* max depenetration velocity (unitree.py:33 -> 1.0) */
  float contact_vstick;   /* |v_t| below which static friction applies */
  float limit_k, limit_c; /* joint-limit spring / damper */
  float force_threshold;  /* contact sensor threshold, 1.0 N [UPSTREAM ContactSensorCfg] */
  float self_k;           /* self-collision penalty stiffness N/m (explicit: keep self_k dt^2 well below the lightest link's mass) */
} rl_sim_desc;

/* ---- terrain ------------------------------------------------------------------------------ */
typedef struct rl_terrain_desc {
  int32_t is_plane;       /* flat_env_cfg.py:18-19 */
  int32_t nx, ny;         /* heightfield samples (x-major: h[ix*ny+iy]) */
  float hscale;           /* sample spacing */
  float x0, y0;           /* world coords of sample (0,0) */
  int32_t num_rows, num_cols; /* sub-terrain grid (levels x types) */
  float tile_size;        /* 8 m */
  float border;           /* 20 m */
  int32_t max_init_level; /* velocity_env_cfg.py:51 */
  int32_t curriculum;     /* terrain_levels_vel enabled (velocity_env_cfg.py:671) */
} rl_terrain_desc;

/* ---- task: term stack of ManagerBasedRLEnv.step() ---------------------------------------- */
typedef struct rl_task_desc {
  float episode_length_s; /* velocity_env_cfg.py:715 */
  /* commands: velocity_env_cfg.py:106-117, VEL/mdp/commands.py:22-92 */
  float cmd_range[4][2];  /* vx, vy, wz, heading */
  float cmd_resample[2];
  float cmd_rel_standing, cmd_rel_heading, cmd_heading_stiffness;
  int32_t cmd_heading;
  float cmd_small_threshold; /* commands.py:47 -> 0.2 */
  /* observations */
  int32_t n_policy, n_critic;
  rl_obs_term policy[RL_MAX_OBS_TERMS], critic[RL_MAX_OBS_TERMS];
  int32_t policy_corrupt, critic_corrupt;
  int32_t scan_nx, scan_ny; float scan_res; float scan_offset; /* 17 x 11 @0.1, offset 0.5 */
  int32_t scan_body;      /* body the RayCaster rides on (velocity_env_cfg.py:71 prim_path; G1: torso_link, unitree_g1/rough_env_cfg.py:55) */
  uint32_t wheel_joint_mask;
  /* rewards */
  int32_t n_rewards;
  rl_reward_term rewards[RL_MAX_REWARD_TERMS];
  /* terminations: velocity_env_cfg.py:648-664 */
  int32_t term_time_out, term_out_of_bounds, term_illegal_contact;
  float oob_buffer; uint64_t illegal_body_mask; float illegal_threshold;
  /* events: velocity_env_cfg.py:262-371 */
  int32_t ev_material, ev_mass_base, ev_mass_others, ev_com, ev_wrench, ev_reset_joints, ev_gains, ev_reset_base, ev_push;
  float friction_static[2], friction_dynamic[2], restitution[2]; int32_t friction_buckets;
  float mass_base_add[2]; uint64_t mass_base_mask;
  float mass_scale[2]; uint64_t mass_scale_mask;
  float com_range[3][2]; uint64_t com_mask;
  float wrench_force[2], wrench_torque[2];
  float reset_joint_pos_scale[2], reset_joint_vel_scale[2];
  float gain_kp_scale[2], gain_kd_scale[2];
  float reset_pose[6][2], reset_vel[6][2];
  float push_interval[2], push_vel[6][2];
  int32_t base_body;      /* body addressed by the mass-add / COM / external-wrench events (base_link_name; G1: torso_link) */
  /* curriculum: command_levels_lin_vel / command_levels_ang_vel (VEL/mdp/curriculums.py:21-94; velocity_env_cfg.py:673-690).
   * The command ranges of lin_vel_x / lin_vel_y (resp. ang_vel_z) start at range * mult[0] and, whenever the step counter is
   * a multiple of the episode length, widen by 0.1 on either side (clamped to range * mult[1]) if the mean episode sum of
   * reward term `*_term` over the envs reset in that step, divided by the episode length in seconds, exceeds 0.8 x its weight.
   * Deviation from the reference: the widened range takes effect from the NEXT step (there it already applies to the commands
   * resampled by that very reset): the decision needs a reduction over all envs of the step, i.e. the end of the launch. */
  int32_t cur_cmd_lin, cur_cmd_ang;            /* term enabled */
  int32_t cur_cmd_lin_term, cur_cmd_ang_term;  /* reward term index (reward_term_name) */
  float cur_cmd_lin_mult[2], cur_cmd_ang_mult[2]; /* range_multiplier */
} rl_task_desc;

typedef struct rl_env_desc {
  rl_model_desc model;
  rl_sim_desc sim;
  rl_terrain_desc terrain;
  rl_task_desc task;
} rl_env_desc;

/* ---- buffers a caller may look at (device pointers) -------------------------------------- */
enum rl_buffer {
  RL_BUF_OBS_POLICY = 0,   /* float [N, obs_policy_dim]: the buffer the LAST step()/reset() wrote (slot rl_env_obs_slot of the ring below) */
  RL_BUF_OBS_CRITIC = 1,   /* float [N, obs_critic_dim]: likewise */
  RL_BUF_REWARD = 2,       /* float [N] */
  RL_BUF_TERMINATED = 3,   /* uint8 [N] */
  RL_BUF_TIME_OUT = 4,     /* uint8 [N] */
  RL_BUF_EPISODE_LENGTH = 5, /* int64 [N]  (settable: rsl_rl init_at_random_ep_len, train.py:224) */
  RL_BUF_ROOT_STATE = 6,   /* float [N, 13] pos(3) quat wxyz(4) lin vel(3) ang vel(3), world; written by rl_env_export_state */
  RL_BUF_JOINT_POS = 7,    /* float [N, D] */
  RL_BUF_JOINT_VEL = 8,    /* float [N, D] */
  RL_BUF_REWARD_TERMS = 9, /* float [T, N] per-term weighted value of the last step (parity checks) */
  RL_BUF_EPISODE_SUMS = 10,/* float [T, N] reward_manager._episode_sums (VEL/mdp/curriculums.py:44) */
  RL_BUF_COMMAND = 11,     /* float [N, 3] */
  RL_BUF_CONTACT_FORCE = 12, /* float [N, B, 3] net_forces_w of the last substep (inspection view: see rl_env_get_buffer) */
  RL_BUF_CONTACT_TIMERS = 13, /* float [N, B, 4] current_air, current_contact, last_air, last_contact */
  RL_BUF_LOG = 14,         /* float [RL_LOG_RING][RL_LOG_PARTS][RL_LOG_SIZE] device-side episode log: step k accumulates into slot
                              k % RL_LOG_RING, which step k - 1 zeroed (see rl_env_log_slot).  A slot is RL_LOG_PARTS partial rows
                              - a reader SUMS them (the wavefronts of a launch spread their atomic adds over the rows: adds on one
                              address are serialised, and a launch in which a third of the envs reset would spend half its time
                              there).  Step k + 1 then lets a slot whose step reset nobody inherit slot k - 1 row by row, so every
                              slot but the newest reads as "the log of the most recent step that reset an env"; for the newest, a
                              reader does that select itself: summed entry 0 (the reset count) > 0 ? slot k : slot k - 1.  Entry
                              RL_LOG_SIZE - 1 counts the resets of the slot's own step (not inherited) */
  RL_BUF_ACTION = 15,      /* float [N, A] last (raw) action */
  RL_BUF_JOINT_TORQUE = 16,/* float [N, D] applied torque of the last substep */
  RL_BUF_JOINT_ACC = 17,   /* float [N, D] */
  RL_BUF_ENV_ORIGIN = 18,  /* float [N, 3] */
  RL_BUF_TERRAIN_LEVEL = 19, /* int32 [N] */
  RL_BUF_TASK_STATE = 20,  /* float [N, RL_TASK_STATE_NF] command / event state carried between steps (rl_task_state_field);
                              written by rl_env_export_state, read by rl_env_commit_state */
  RL_BUF_GAINS = 21,       /* float [N, 2, D] per-env actuator stiffness / damping (randomize_actuator_gains, velocity_env_cfg.py:337-347) */
  RL_BUF_OBS_POLICY_RING = 22, /* float [2, Npad, obs_policy_dim] both observation buffers; step()/reset() alternate between them */
  RL_BUF_OBS_CRITIC_RING = 23, /* float [2, Npad, obs_critic_dim] */
  RL_BUF_CMD_LEVELS = 24,  /* float [16]: current command ranges of the command_levels_* curricula - lin_vel_x lo/hi, lin_vel_y lo/hi,
                              ang_vel_z lo/hi (what the reference logs as Curriculum/command_levels_lin_vel = [1], _ang_vel = [5]),
                              then the accumulators of the running decision (sum, count per term) */
  RL_BUF_COUNT
};

/* fields of one RL_BUF_TASK_STATE row: UniformVelocityCommand state [UPSTREAM B7] (command, heading target, resampling
 * timer, the two metric accumulators, the two per-env flags), the interval-event timer of push_robot [UPSTREAM B2] and the
 * persistent external wrench of apply_external_force_torque [UPSTREAM B8] */
enum rl_task_state_field {
  RL_TS_CMD_VX = 0, RL_TS_CMD_VY, RL_TS_CMD_WZ, RL_TS_HEADING_TARGET, RL_TS_CMD_TIME_LEFT, RL_TS_METRIC_XY, RL_TS_METRIC_YAW,
  RL_TS_PUSH_TIME_LEFT, RL_TS_IS_HEADING_ENV, RL_TS_IS_STANDING_ENV, RL_TS_EXT_FORCE, RL_TS_EXT_TORQUE = RL_TS_EXT_FORCE + 3,
  RL_TASK_STATE_NF = RL_TS_EXT_TORQUE + 3
};

#define RL_LOG_SIZE 64
#define RL_LOG_PARTS 32 /* partial rows of a ring slot, summed by the reader */
#define RL_LOG_RING 64 /* a step's log stays readable until RL_LOG_RING - 2 further steps have been launched */

typedef struct rl_env rl_env; /* opaque */

/* Replaces: ManagerBasedRLEnv.__init__(cfg) [UPSTREAM ctor; call site train.py:177].
 * HOST arrays: `terrain_heights` nx*ny floats and `terrain_origins` [num_rows][num_cols][3]
 * (both NULL for a plane), `env_origins` [num_envs][3] (plane only, else NULL).  `device` is the
 * HIP device ordinal.  Runs the "startup" events (velocity_env_cfg.py:262-314). */
int rl_env_create(const rl_env_desc* desc, const float* terrain_heights, const float* terrain_origins,
                  const float* env_origins, int32_t num_envs, uint64_t seed, int32_t device, rl_env** out);

/* Replaces: ManagerBasedRLEnv.reset() -> _reset_idx(all) + observation compute.
 * env_ids == NULL resets every env. env_ids is a HOST array. */
int rl_env_reset(rl_env* env, const int32_t* env_ids, int32_t n, void* stream);

/* Replaces: ManagerBasedRLEnv.step(action).  `action_dev` is a device pointer to
 * float [N, A] row-major.  Stream-ordered; no host synchronisation. */
int rl_env_step(rl_env* env, const float* action_dev, void* stream);

/* rl_env_step fused with what rsl_rl's PPO.process_env_step stores for the transition (train.py:224 -> OnPolicyRunner.learn;
 * include/rl_rollout.h): besides everything rl_env_step does, the same kernel writes
 *     rewards_out[e] = reward[e] + gamma * values[e] * time_out[e]      (bootstrapping on time outs)
 *     dones_out[e]   = terminated[e] | time_out[e]
 * for e < N - normally straight into the current slot of an rl_rollout (rl_rollout_record_slots), which saves the
 * separate record launch.  values_dev: float [N] (the critic's V(s_t)); all three are device pointers.
 * values_dev == NULL defers the bootstrap (the critic of step t is not on the path to env step t: it may still be running on another
 * stream, robot_lab_amd/collect.py): the kernel then writes the RAW reward and  dones_out[e] = (terminated | time_out) | time_out << 1;
 * rl_rollout_compute_returns adds gamma * V(s_t) where bit 1 is set and clears the bit - the storage ends up with the same numbers. */
int rl_env_step_record(rl_env* env, const float* action_dev, const float* values_dev, float* rewards_out_dev, uint8_t* dones_out_dev,
                       float gamma, void* stream);

/* Device pointer + shape of one of the env-owned buffers.  shape[] gets up to 3 dims, ndim out.
 * RL_BUF_CONTACT_FORCE / JOINT_TORQUE / JOINT_ACC are inspection views: they are allocated on the first
 * request and filled by every step() AFTER that request (the training path never pays for them). */
int rl_env_get_buffer(rl_env* env, int32_t which, void** dev_ptr, int64_t shape[3], int32_t* ndim,
                      int32_t* elem_size);

/* Gathers the SoA simulator state into the AoS debug/inspection buffers.  Not part of step(); callers such as
 * rl_utils.py:12-13 (camera follow) use it.  The set it writes is everything step() carries from one call to the next
 * besides the directly addressable buffers (EPISODE_LENGTH, EPISODE_SUMS, TERRAIN_LEVEL): ROOT_STATE, JOINT_POS, JOINT_VEL,
 * ACTION (= last action), GAINS, CONTACT_TIMERS, TASK_STATE, ENV_ORIGIN. */
int rl_env_export_state(rl_env* env, void* stream);

/* The inverse of rl_env_export_state: scatters those AoS device buffers back into the simulator state (the reference's
 * `scene.write_data_to_sim()` / `write_root_state_to_sim` / `write_joint_state_to_sim` direction [UPSTREAM B1]).  A caller edits
 * the buffers (they are plain device pointers) after an export and commits.  Used by the teacher-forced parity tests
 * (tests/test_gpu_teacher_forced.py): HIP and oracle are stepped ONCE from a shared state. */
int rl_env_commit_state(rl_env* env, void* stream);

/* Convenience form for HOST arrays: export, overwrite the given parts (any pointer may be NULL to leave that part as is;
 * root_state [N,13], joint_pos [N,D], joint_vel [N,D]), commit.  Synchronises `stream`. */
int rl_env_import_state(rl_env* env, const float* root_state, const float* joint_pos,
                        const float* joint_vel, void* stream);

/* Copies the RL_LOG_SIZE episode-log accumulators of the most recent step that reset an environment to host (what a caller of
 * the reference finds in extras["log"], which is rebuilt inside _reset_idx only: manager_based_rl_env.py [UPSTREAM B1]).
 * Synchronises `stream`. */
int rl_env_read_log(rl_env* env, float* out_host, void* stream);
/* Ring slot of RL_BUF_LOG the last step() wrote (= steps so far % RL_LOG_RING); device readers use it to avoid the copy. */
int32_t rl_env_log_slot(const rl_env* env);

/* Which of the two observation buffers (RL_BUF_OBS_*_RING) the last step()/reset() wrote: 0 or 1. */
int32_t rl_env_obs_slot(const rl_env* env);
/* ManagerBasedRLEnv.common_step_counter [UPSTREAM B1]: steps taken so far.  It keys the counter-based random streams
 * (resets, command resampling, pushes, observation noise), so a state moved between two envs moves together with it. */
int64_t rl_env_step_count(const rl_env* env);
int rl_env_set_step_count(rl_env* env, int64_t count);

/* ---- hipGraph capture of a loop around rl_env_step (the 24-step collection loop of train.py:224 -> OnPolicyRunner.learn) --------
 * Everything a step launch takes from the host is either a pointer that repeats with period 2 (the observation buffers) or the
 * step count, and the kernels read the step count as  *device word + launch literal.  So a stream capture that contains an EVEN
 * number n of step launches replays correctly any number of times:
 *     rl_env_graph_begin(env, stream);            before hipStreamBeginCapture: anchors the device word at the current count
 *     hipStreamBeginCapture(stream, ...);  n x { ...; rl_env_step[_record](env, ..., stream); ... }
 *     rl_env_graph_end(env, stream);              still inside the capture: appends the node that advances the device word by n,
 *                                                 rolls the host-side mirrors back (capturing ran nothing); returns n or -1
 *     hipStreamEndCapture(stream, &graph); hipGraphInstantiate(...);
 *     per replay:  rl_env_graph_launching(env, stream);  hipGraphLaunch(exec, stream);
 * rl_env_graph_launching accounts the n steps on the host side (rl_env_step_count, rl_env_log_slot, rl_env_obs_slot) and fails
 * when the env is not where the capture found it (observation slot parity).  Direct rl_env_step calls may be mixed with replays. */
int rl_env_graph_begin(rl_env* env, void* stream);
int rl_env_graph_end(rl_env* env, void* stream);
int rl_env_graph_launching(rl_env* env, void* stream);

int32_t rl_env_num_envs(const rl_env* env);
int32_t rl_env_num_actions(const rl_env* env);
int32_t rl_env_obs_dim(const rl_env* env, int32_t group); /* 0 policy, 1 critic */
int32_t rl_env_max_episode_length(const rl_env* env);
/* Environments a 64-lane wavefront simulates: 4 (sixteen lanes per env - the latency mapping, what <= ~8 k quadruped envs per GPU and
 * every trunk + limbs robot get) or 16 (one lane per limb - the throughput mapping of large launches).  Chosen by rl_env_create from
 * the launch size (csrc/rl_env.hip envs_per_wave; RL_ENV_SUB=4|1 forces either); results do not depend on it beyond fp32 round-off.
 * No counterpart in the reference (PhysX picks its own launch geometry): informational. */
int32_t rl_env_envs_per_wavefront(const rl_env* env);
/* The step kernel this env runs: 0 = the term-stack interpreter (reward / observation descriptors read from the table image - every
 * task), > 0 = the id of a kernel SPECIALISED on the task (csrc/env_spec.h: the task's reward terms as compile-time constants), picked
 * by rl_env_create when the compiled tables equal, bit for bit, the constants that kernel was generated from (RL_ENV_SPEC=0: never).
 * Same results up to fp32 summation order.  No counterpart in the reference: informational. */
int32_t rl_env_spec_id(const rl_env* env);

/* ---- specialising ANY task at run time (robot_lab_amd/jit.py drives these; nothing of it in the reference) ---------------------------
 * The library carries specialised step kernels for eight tasks (csrc/spec/).  Any other task - or an edited cfg - can get its own:
 *   rl_env_spec_source   the C++ source of the task's Spec (`struct <struct_name>`, spec id `id` >= 1000) from its descriptor, through the
 *                        same descriptor -> tables compile rl_env_create runs.  Host code, no device.  Returns the length written into
 *                        `out` (0: the task cannot be specialised - rl_env_last_error says why; -1: `cap` too small).
 *   (the caller compiles it with hipcc against this library's own csrc headers into a shared object exporting rl_spec_plugin_abi /
 *    _id / _matches / _launch: robot_lab_amd/jit.py has the five-line translation unit)
 *   rl_env_register_spec_plugin   dlopens that object and adds it to the process-wide list rl_env_create tries after the built-in Specs;
 *                        refused unless it was compiled against the headers this library was (rl_env_abi_stamp: a digest of csrc/).
 * An env created afterwards from tables that equal the plugin's constants bit for bit runs its kernels: rl_env_spec_id() = `id`. */
int rl_env_spec_source(const rl_env_desc* desc, const char* struct_name, const char* task, int id, char* out, int cap);
int rl_env_register_spec_plugin(const char* so_path);
int32_t rl_env_spec_plugin_count(void);
const char* rl_env_abi_stamp(void);

/* The launch geometry rl_env_create + rl_env_step arrive at for `num_envs` environments of this task on a device with `n_cu` compute
 * units (<= 0: 256, MI355X), without creating anything - no device is touched:
 *   out[0] lanes per limb (4 / 2 / 1 quadrupeds, 8 / 4 trunk + limbs), out[1] wavefronts per workgroup of the step launch (4 or 1),
 *   out[2] LDS bytes of a single-wavefront workgroup (table image + one wavefront's region), out[3] LDS bytes of the launched workgroup.
 * Fails (as rl_env_create would) when no lane mapping fits a CU's 160 KiB of LDS.  Informational - tests/test_launch_plan.py pins the
 * choice for the BASELINE robots with it, so that table growth which pushes a mapping off a CU fails a test, not a benchmark sweep. */
int rl_env_plan(const rl_env_desc* desc, int32_t num_envs, int32_t n_cu, int32_t out[4]);

int rl_env_destroy(rl_env* env);
const char* rl_env_last_error(void);
/* ABI guard: sizeof(rl_env_desc) the library was built with. */
uint64_t rl_env_desc_size(void);

#ifdef __cplusplus
}
#endif
#endif /* RL_ENV_H_ */

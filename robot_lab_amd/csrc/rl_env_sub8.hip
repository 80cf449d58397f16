// rl_env_sub8.hip - translation unit of the env kernels with eight lanes per limb (trunk + limbs instances, 2 envs per wavefront); see rl_env_kernels.h
#include "rl_env_kernels.h"
#define RL_ENV_TU_SUB 8
#include "rl_env_sub.inl"

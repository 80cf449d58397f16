// rl_env_sub2.hip - translation unit of the env kernels with a lane pair per limb (8 envs per wavefront); see rl_env_kernels.h
#include "rl_env_kernels.h"
#define RL_ENV_TU_SUB 2
#include "rl_env_sub.inl"

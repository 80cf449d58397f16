// rl_env_sub1.hip - translation unit of the env kernels with one lane per limb (16 envs per wavefront); see rl_env_kernels.h
#include "rl_env_kernels.h"
#define RL_ENV_TU_SUB 1
#include "rl_env_sub.inl"

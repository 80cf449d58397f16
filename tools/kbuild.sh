#!/bin/bash
# Build ONE lane-program instance of the env kernel into a scratch directory and print its resource usage + static
# instruction profile:   tools/kbuild.sh <outdir> <CL*10+SUB, e.g. 34 | 44 | 74> [extra hipcc flags]
OUT=$1; INST=${2:-34}; shift; shift
mkdir -p $OUT && cd $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-remove-redundant-endcf=false -std=c++17 -shared -fPIC -DRL_ENV_SINGLE_TU -DRL_ENV_ONLY=$INST -DRL_PHASE_MARKS -save-temps -o lib_marks.so /root/repo/robot_lab_amd/csrc/rl_env.hip "$@" 2>&1 | grep -E "error" -A6 | head -40
python - <<PY
import re
s=open('$OUT/rl_env-hip-amdgcn-amd-amdhsa-gfx950.s').read()
# the step kernels (RESET = 0): template <Topo, RESET, SUB, WGW>; the last one written to step.s is the four-wavefront-workgroup variant when it exists
for m in sorted(re.finditer(r'^(_ZN12_GLOBAL__N_110env_kernel[^:\n]*Li0ELi\dELi(\d)EEEvNS1_6KStateEPKvj):', s, flags=re.M), key=lambda m: int(m.group(2))):
    i=m.start(); j=s.index('s_endpgm', i)
    open('$OUT/step.s','w').write(s[i:j+10])
    tail=s[j:j+4000]
    print(m.group(1)[:70], [x for x in re.findall(r'; (NumVgprs: \d+|NumAgprs: \d+|ScratchSize: \d+|codeLenInByte = \d+)', tail)][:4])
PY
python /root/repo/tools/isa_profile.py $OUT/step.s 2>/dev/null | sed -n 1,28p

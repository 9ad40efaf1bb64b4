#!/bin/bash
# lockstep soaks against the f32 CPU oracle at BASELINE's full sizes (one gpurun call; writes gpurun_out/<tag>/soak_parity.txt)
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
{
echo "# tests/soak_parity.py on one MI355X box (the final binaries of the round): lockstep against the f32 CPU oracle at BASELINE's full batch sizes, TOL = 0"
for args in "swing 4096 1040" "swing 16384 520" "swing 32768 260" "swing 1048576 104" "tennis 1048576 300" "swing 4194304 52" "swing 1048576 52 rg" "swing 131072 104 rg" "swing 4096 520 rg" "swing 32768 1040 defer_all" "swing 65536 260 defer" "tennis 65536 600"; do
  timeout -k 10 400 python3 tests/soak_parity.py $args 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
done
for args in "4096 1040" "4096 520 3" "1000 520 1 rg" "20000 104"; do
  timeout -k 10 500 python3 tests/soak_policy.py $args 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
done
timeout -k 10 900 python3 tests/soak_seal.py 12 32768 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
} > $O/soak_parity.txt
cat $O/soak_parity.txt

for cot in 5 10; do echo "== COT $cot"; FSC_L16_COT=$cot python tools/l16_check.py --iters 10 b1e b1c2 b2e 2>&1 | grep -v amdgpu | cut -c1-46,118-220; done
for cot in 8; do echo "== COT $cot"; FSC_L16_COT=$cot python tools/l16_check.py --iters 10 b2e b2c2 2>&1 | grep -v amdgpu | cut -c1-46,118-220; done
for cot in 4 7; do echo "== COT $cot"; FSC_L16_COT=$cot python tools/l16_check.py --iters 10 b0c2 2>&1 | grep -v amdgpu | cut -c1-46,118-220; done

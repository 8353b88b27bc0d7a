#!/bin/bash
# Development aid: registers / scratch of the fused kernel as built by the product flags
P=dh-aug-dh-forward-kinematics-model-driven-augmentation-for-3d-human-pose-estimation_amd
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=fast -fno-signed-zeros -ffinite-math-only -Iinclude -I$P/csrc -mllvm -amdgpu-mfma-vgpr-form $@ -S --cuda-device-only -o /tmp/isa/mlp.s $P/csrc/dhaug_mlp.hip 2>/dev/null
grep -E "^\s+\.(vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size):" /tmp/isa/mlp.s | head -4 | tr -d '\n'; echo; echo "scratch instrs: $(grep -c scratch_ /tmp/isa/mlp.s)  lines: $(wc -l < /tmp/isa/mlp.s)"

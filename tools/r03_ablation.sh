#!/bin/bash
# r03: the loop-body ablation of the fused LK kernel (what staging image tiles in LDS could buy at most) and the streaming
# ceiling of its traffic pattern, on the r03 code, in ONE call (boxes differ by 10-15 %).  -> gpurun_out/r03_fused_ablation.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03_fused_ablation.txt
mkdir -p gpurun_out
{
echo "# fused LK kernel, ESM+SSD+Homography 200x200 x 64 targets, one gpurun call; commit $(cat .git_head 2>/dev/null)"
echo "# variants: complete | NOTEX (no texel fetches: the bound on what an LDS image tile could save) | NOMATH (no sampling arithmetic) |"
echo "#           NOACC (no Gram / gradient accumulation) | all three | TRIVIAL (same loads and stores, one add)"
for extra in "" "-DMTFHIP_EXPERIMENT_NOTEX" "-DMTFHIP_EXPERIMENT_NOMATH" "-DMTFHIP_EXPERIMENT_NOACC" "-DMTFHIP_EXPERIMENT_NOTEX -DMTFHIP_EXPERIMENT_NOMATH -DMTFHIP_EXPERIMENT_NOACC" "-DMTFHIP_EXPERIMENT_TRIVIAL"; do
  echo "## EXTRA='$extra'"
  SWEEP_EXTRA="$extra" bash tools/sweep_fused.sh "1 2 4"
done
echo "## membench_streams (tools/membench_streams.hip): streaming kernel with this kernel's traffic; args: targets dyn_lds interleave nonzero"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench_streams.hip 2>/dev/null
echo "### zeros in the buffers"; /tmp/membench 64 0 0 0 | head -12
echo "### non-zero payload";   /tmp/membench 64 0 0 1 | head -12
} 2>&1 | tee $out

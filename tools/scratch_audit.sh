#!/bin/bash
# Every kernel of the library that uses scratch (register spills), by bytes per lane.  A spill inside a loop that keeps LDS-DMA or
# prefetch loads in flight is reloaded by scratch_load + s_waitcnt vmcnt(0) - a drain of everything outstanding: conv_s2d_kernel
# lost a third of its time that way.  Expected output at the end of round 5: the 128-channel and statistics
# instantiations of conv_s2f_kernel (off the default path), wino_gemm_kernel<128,2> (48 B, reloaded per tile, not in the k-loop),
# wino_wgrad_gemm_kernel<4> and conv_igemm_bf16_multi_kernel.      bash tools/scratch_audit.sh [file.hip ...]
cd "$(dirname "$0")/../mmhand_amd/csrc"
for f in ${@:-*.hip}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm \
      -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_audit.o 2>&1 |
  awk -v F=$f '/Function Name/{name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name)}
               /ScratchSize/{s=$0; sub(/.*ScratchSize \[bytes\/lane\]: /,"",s); sub(/ .*/,"",s); if (s+0>0) print s, F, name}'
done | sort -n -r

#!/bin/bash
# device assembly of the library into /tmp/asm/rp.s and, per kernel given on the command line (mangled prefix), its length, scratch use and register counts
mkdir -p /tmp/asm
cd /root/repo/roboticsplayroompybullet_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -DRP_BUILD_ID=\"x\" $KASM_FLAGS --cuda-device-only -S -o /tmp/asm/rp.s rp_playroom.hip 2>&1 | grep -E "error" -A3 | head -20
for k in "$@"; do
  awk "/^$k/,/s_endpgm/" /tmp/asm/rp.s > /tmp/asm/$k.s
  echo "$k: $(wc -l < /tmp/asm/$k.s) lines, scratch ops $(grep -c scratch_ /tmp/asm/$k.s), $(grep -A30 "\.name:           $k" /tmp/asm/rp.s | grep -E "vgpr_count|private_segment_fixed|group_segment_fixed" | tr -s ' ' | tr '\n' ' ')"
done

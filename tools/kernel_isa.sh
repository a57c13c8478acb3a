#!/bin/bash
# usage: tools/kernel_isa.sh <kernel-name-substring>  -> build/<name>.s + instruction histogram + resources
set -e
cd /root/repo
make -C gorder_amd/csrc asm >/dev/null 2>&1
cd build
S=gorder_hip-hip-amdgcn-amd-amdhsa-gfx950.s
sym=$(grep -oE "^_Z[A-Za-z0-9_]*$1[A-Za-z0-9_]*:" $S | head -1 | tr -d ':')
a=$(grep -n "^$sym:" $S | cut -d: -f1); b=$(grep -n "amdhsa_kernel $sym" $S | cut -d: -f1)
sed -n ${a},${b}p $S > $1.s
echo "$sym: $(wc -l < $1.s) lines"
grep -oE "^\s+(v_|s_|ds_|global_|buffer_|flat_)[a-z0-9_]+" $1.s | sort | uniq -c | sort -rn | head -${2:-25}
grep -A10 "Function Name: $sym" resource_usage.txt | grep -E "VGPRs:|SGPRs:|Occupancy|Spill|LDS"

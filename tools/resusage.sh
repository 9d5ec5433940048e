#!/bin/bash
# usage: tools/resusage.sh file.hip [grep-filter]   -> one line per kernel: name sgpr vgpr agpr occ lds
f=$1; flt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I /root/repo/include -c $f -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | awk '/Function Name/{if(n)print n,s,v,a,o,l,sp; n=$3} /TotalSGPRs/{s="sgpr="$2} /^VGPRs:/{v="vgpr="$2} /AGPRs/{a="agpr="$2} /Occupancy/{o="occ="$4} /LDS Size/{l="lds="$5} /VGPRs Spill/{sp="spill="$3} END{print n,s,v,a,o,l,sp}' | c++filt | grep -E "$flt"

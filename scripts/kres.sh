#!/bin/bash
# kernel resource usage of one csrc file:  scripts/kres.sh raster_fwd.hip
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -Rpass-analysis=kernel-resource-usage \
  /root/repo/handobjectconsist_amd/csrc/$1 -o /tmp/kres.o 2>&1 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | \
  awk '/Function Name/ {if (line) print line; line=$3} /TotalSGPRs|VGPRs:|Spill|Occupancy|LDS Size|ScratchSize/ {line=line" | "$0} END {print line}'

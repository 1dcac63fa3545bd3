#!/bin/bash
# Developer tool: register / scratch / LDS usage of the kernels of one source file (hipcc remarks), e.g.
#   tools/kres.sh scp_qp_persist16.hip [grep pattern]
cd "$(dirname "$0")/../ba-path-planning_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I../../include -I. ${KRES_FLAGS:-} \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  grep -E "error|Name:|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' |
  awk '/Name:/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' | grep -E "${2:-.}"

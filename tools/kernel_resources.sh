#!/bin/bash
# registers / scratch / LDS / occupancy of every kernel of a translation unit:  tools/kernel_resources.sh mkv_update.hip
cd "$(dirname "$0")/../megakv_amd/csrc"
case "$1" in mkv_update.hip) EXTRA="$EXTRA -mllvm -disable-machine-licm";; esac    # (as the Makefile builds it)
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-exceptions -fno-rtti -I../../include -DMEM_P=30 -DHASH_CUCKOO=1 $EXTRA \
	-c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
	grep -E "Function Name|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' |
	paste - - - - - - | c++filt | awk -F'\t' '{n=$1; sub(/Function Name: /,"",n); sub(/\(.*/,"",n); printf "%-60s %s | %s | %s | %s | %s\n", substr(n,1,60), $2,$3,$4,$5,$6}'

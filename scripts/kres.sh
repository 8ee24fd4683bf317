#!/bin/bash
# registers / spills per kernel of one source file:  scripts/kres.sh use_conv_v4 [extra flags]
F=$1; shift
cd $(dirname $0)/../universal_speech_enhancement_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -c $F.hip -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - | sed 's/Function Name: //'

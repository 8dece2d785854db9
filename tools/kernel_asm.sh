#!/bin/bash
# Device assembly of one csrc/*.hip with the flags of mici_amd/build.py:  tools/kernel_asm.sh k_implicit_blk16 [-DMM_DEV_KERNELS] -> /tmp/<name>.s
set -e
name=${1%.hip}; shift
cd "$(dirname "$0")/../mici_amd/csrc"
extra=""
case $name in k_implicit_mfma|k_implicit_blk16|k_implicit_pair|k_implicit_fork) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=on $extra "$@" --cuda-device-only -S $name.hip -o /tmp/$name.s
echo /tmp/$name.s

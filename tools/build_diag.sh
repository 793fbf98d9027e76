#!/bin/bash
# Diagnostic build: libtedm_diag.so = the product sources + -DTEDM_DIAG_TRACE (in-kernel s_memtime phase stamps in
# conv_igemm_kernel, dumped by tedm_op_conv2d when TEDM_TRACE_FILE is set).  Never loaded by the product path.
set -e
cd "$(dirname "$0")/../tedm_amd/csrc"
mkdir -p ../build/diag
objs=()
for f in conv_igemm_p0 conv_igemm_p1 conv_igemm_p2 conv_igemm_p3 conv_igemm_p4 conv_igemm pack elementwise attention linattn_fused wgrad wgrad_mfma backward head_train attention_bwd unet_host; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DTEDM_DIAG_TRACE -c $f.hip -o ../build/diag/$f.o &
  objs+=(../build/diag/$f.o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtedm_diag.so "${objs[@]}"
echo built tedm_amd/libtedm_diag.so

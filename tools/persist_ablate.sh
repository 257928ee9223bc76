#!/bin/bash
# Ablation builds of the persistent GEMM (gemm_persist.h, MMF_PERSIST_ABL): libmmf_amd.pabl1.so (no epilogue stores), libmmf_amd.pabl2.so (K-loops only).
# Run HERE (cross-compiles); on the GPU box:  MMF_AMD_LIB=mmf_amd/libmmf_amd.pabl1.so python tools/gemm_ab.py --tun 18:1,2,3
for a in 1 2; do
  MMF_AMD_EXTRA_HIPCC_FLAGS="-DMMF_PERSIST_ABL=$a" python -m mmf_amd.csrc.build --tag pabl$a
done

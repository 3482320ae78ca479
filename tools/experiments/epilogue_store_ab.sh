#!/bin/bash
# How much of the persistent GEMM's epilogue is the C stores reaching memory?  Twin build with -DV7_LAB_DROP_STORES (zero-length
# C descriptors: the stores are issued and dropped) against the product library, same box, training shapes.
#   make -C visitron_amd/csrc OUT_DIR=../lib_lab OBJ_DIR=../../build/csrc_lab "CXXFLAGS=... -DV7_LAB_DROP_STORES=1"
set -e
mkdir -p gpurun_out/epi
O=gpurun_out/epi/store_ab.txt
: > $O
for shape in "50820 2304 768 0 0" "50820 768 768 0 0" "50820 3072 768 1 1" "50820 768 3072 0 0" "50820 768 2304 0 0"; do
  python3 tools/epilogue_ab.py $shape 16 >> $O
  VT_HIP_LIB=$PWD/visitron_amd/lib_lab/libvisitron_hip.so python3 tools/epilogue_ab.py $shape 16 >> $O
done
cat $O

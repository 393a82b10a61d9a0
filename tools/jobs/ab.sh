mkdir -p gpurun_out/ab
for v in "$@"; do AB_CASES=${AB_CASES:-g1_8,g1_12} MRCZ_LIB_PATH=$PWD/datacompressionfloat_amd/lib/ab/$v.so python tools/ab_codec.py $v 2>&1 | grep -v amdgpu.ids; done

set -e
R=$(pwd); B=$R/datacompressionfloat_amd/bin
D=/dev/shm/mrcz_cli_$$; mkdir -p $D
python3 - <<PY
import numpy as np
rng=np.random.default_rng(1)
x=rng.normal(10,3,1<<28).astype(np.float32)
x[:256]=0
x.tofile("$D/vol.mrc")
PY
export MRCZ_TRACE=1
t() { local a=$(date +%s.%N); "$@" > $D/o.log 2> $D/e.err; local b=$(date +%s.%N); echo "wall $(python3 -c "print(round($b-$a,3))") s: $1 $6 $7"; cat $D/e.err | grep -v amdgpu.ids | tail -8; }
for i in 1 2 3; do t $B/mrc_tar -i $D/vol.mrc -o $D/vol.zip -t zip -b 8; done
for i in 1 2; do t $B/mrc_tar -i $D/vol.zip -o $D/vol.out -t unzip; done
export MRCZ_NO_MMAP=1
t $B/mrc_tar -i $D/vol.mrc -o $D/vol.zip -t zip -b 8
rm -rf $D

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
$B/mrc_tar -i $D/vol.mrc -o $D/vol.zip -t zip -b 8 > /dev/null 2>&1
t() { local a=$(date +%s.%N); "$@" > $D/o.log 2> $D/e.err; local b=$(date +%s.%N); echo "wall $(python3 -c "print(round($b-$a,3))") s: $6 $7 fillers=${MRCZ_FILLERS:-3}"; grep -E "run_|main:" $D/e.err | cut -c1-330; }
for f in 3 0 1 2 3 0; do export MRCZ_FILLERS=$f; rm -f $D/vol.out; t $B/mrc_tar -i $D/vol.zip -o $D/vol.out -t unzip; done
for f in 3 0; do export MRCZ_FILLERS=$f; rm -f $D/vol.zip2; t $B/mrc_tar -i $D/vol.mrc -o $D/vol.zip2 -t zip -b 8; done
rm -rf $D

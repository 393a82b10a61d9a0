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
t() { local a=$(date +%s.%N); "$@" > $D/o.log 2> $D/e.err; local b=$(date +%s.%N); echo "wall $(python3 -c "print(round($b-$a,3))") s: $6 $7 teardown=${MRCZ_FULL_TEARDOWN:-0}"; }
for i in 1 2 3 4; do rm -f $D/vol.zip; t $B/mrc_tar -i $D/vol.mrc -o $D/vol.zip -t zip -b 8; done
cp $D/vol.zip $D/vol.fastexit.zip
export MRCZ_FULL_TEARDOWN=1
for i in 1 2 3 4; do rm -f $D/vol.zip; t $B/mrc_tar -i $D/vol.mrc -o $D/vol.zip -t zip -b 8; done
cmp $D/vol.zip $D/vol.fastexit.zip && echo "zip: _exit path == full-teardown path"
unset MRCZ_FULL_TEARDOWN
for i in 1 2 3; do rm -f $D/vol.out; t $B/mrc_tar -i $D/vol.zip -o $D/vol.out -t unzip; done
$B/erasebytes -i $D/vol.mrc -o $D/vol.erased -b 8 > /dev/null
cmp $D/vol.out $D/vol.erased && echo "unzip (_exit path) == erasebytes(input)"
export MRCZ_FULL_TEARDOWN=1
for i in 1 2 3; do rm -f $D/vol.out; t $B/mrc_tar -i $D/vol.zip -o $D/vol.out -t unzip; done
cmp $D/vol.out $D/vol.erased && echo "unzip (full-teardown path) == erasebytes(input)"
rm -rf $D

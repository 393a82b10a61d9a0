# mrc_tarx 8 x 256 MiB, 4 threads, with MRCZ_TRACE: where the wall time goes
set -e
R=$(pwd); B=$R/datacompressionfloat_amd/bin
D=/dev/shm/mrcz_tarx_$$; mkdir -p $D/z $D/u
python3 - <<PY
import numpy as np
for i in range(8):
    rng=np.random.default_rng(100+i)
    x=rng.normal(10,3,1<<26).astype(np.float32); x[:256]=0
    x.tofile("$D/part%d.mrc" % i)
open("$D/files.txt","w").write("\n".join("$D/part%d.mrc" % i for i in range(8))+"\n")
PY
for rep in 1 2; do
  rm -f $D/z/* $D/u/*
  a=$(date +%s.%N); MRCZ_TRACE=1 $B/mrc_tarx -i $D/files.txt -t zip -o $D/z -b 8 -n 4 > $D/zip.out 2> $D/zip.err; b=$(date +%s.%N)
  ls $D/z/* > $D/zips.txt
  MRCZ_TRACE=1 $B/mrc_tarx -i $D/zips.txt -t unzip -o $D/u -n 4 > $D/unzip.out 2> $D/unzip.err; c=$(date +%s.%N)
  python3 -c "print('zip %.3f s  unzip %.3f s' % ($b-$a, $c-$b))"
  echo "--- zip trace"; grep -v "compress .* floats\|uncompress" $D/zip.err | head -40
  echo "--- unzip trace"; grep -v "compress .* floats\|uncompress" $D/unzip.err | head -40
done
rm -rf $D

# mrc_tarx 8 x 256 MiB in /dev/shm with 1, 2, 4, 8 worker threads: wall time of zip and unzip (three repetitions each)
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
for rep in 1 2 3; do for n in 1 2 4 8; do
  rm -f $D/z/* $D/u/*
  a=$(date +%s.%N); $B/mrc_tarx -i $D/files.txt -t zip -o $D/z -b 8 -n $n > /dev/null 2>&1; b=$(date +%s.%N)
  ls $D/z/* > $D/zips.txt
  $B/mrc_tarx -i $D/zips.txt -t unzip -o $D/u -n $n > /dev/null 2>&1; c=$(date +%s.%N)
  python3 -c "print('threads $n: zip %.2f GB/s  unzip %.2f GB/s' % (2.147/($b-$a), 2.147/($c-$b)))"
done; done
rm -rf $D

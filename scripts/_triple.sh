cd /root/repo
for v in tile m32 m32y8 m64c64; do TAUENG_LIB=build_var/$v/libtaueng.so python scripts/march_vs_tile_detail.py child 200,136,24 /tmp/$v.npy 12 2>/dev/null; done
python scripts/march_vs_tile_detail.py child 200,136,24 /tmp/intree.npy 12 2>/dev/null
python - <<'PY'
import numpy as np
n=["tile","m32","m32y8","m64c64","intree"]
a={i:np.load(f"/tmp/{i}.npy") for i in n}
for i in n:
    print(i, [int((a[i]!=a[j]).sum()) for j in n])
PY

cd /root/repo
for rep in 1 2; do for v in tile m32 m64c256 m64c64 intree; do
  if [ $v = intree ]; then unset TAUENG_LIB; else export TAUENG_LIB=build_var/$v/libtaueng.so; fi
  python bench.py --no-configs --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=j['roofline']['kernels']
print('$v', j['value'], k[0]['avg_launch_ms'], k[1]['avg_launch_ms'])"
done; done

echo "allowed cpus: $(grep Cpus_allowed_list /proc/self/status)"; echo "mems: $(grep Mems_allowed_list /proc/self/status)"
lscpu | grep -E "NUMA|L3|Socket|Core"
for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)"; done
echo "L3 domains (first 24):"; for c in $(seq 0 8 184); do echo "cpu$c: $(cat /sys/devices/system/cpu/cpu$c/cache/index3/shared_cpu_list)"; done
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null)"; done
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpu.max
python3 - <<'PY'
import os, threading, time
print("main thread cpu", os.sched_getcpu(), "affinity size", len(os.sched_getaffinity(0)))
cpus=[]
def w():
    t=time.time()
    while time.time()-t<0.05: pass
    cpus.append(os.sched_getcpu())
ts=[threading.Thread(target=w) for _ in range(9)]
[t.start() for t in ts]; [t.join() for t in ts]
print("spinning python threads ran on", cpus)
PY

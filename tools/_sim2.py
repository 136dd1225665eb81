import sys
sys.path.insert(0,'/root/repo/tools')
import sim_depth_bound as s
for stride in (4, 16):
    a=b=0
    for seed in range(4):
        n0,n1 = s.run(seed=seed, stride=stride)
        a+=n0;b+=n1
    print("stride",stride,"ratio",b/a)

import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as bc
which = sys.argv[1]
if which == "D":
    bc.run("cfg-D d=1024 6+6 T=32 S=40", 1024, 6, 6, 32, 40)
else:
    bc.run("shipped d=768 1+3 T=12 S=20", 768, 1, 3, 12, 20)

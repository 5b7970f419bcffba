import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import test_dist_gpu as T
if __name__ == "__main__":
    try:
        T._run_ranks(4, "float32", False, "eager"); print("OK")
    except AssertionError as e:
        print("FAIL", str(e)[:1500], flush=True)

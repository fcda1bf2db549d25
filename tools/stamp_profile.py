"""Copy a measurement JSON into profiles/ with the git commit of the tree it was taken from: tools/stamp_profile.py SRC DST"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = sys.argv[1], sys.argv[2]
text = open(src).read().strip()
d = json.loads(text.splitlines()[-1] if text.count("\n") and not text.lstrip().startswith("{\n") else text)
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "raytracer_amd", "include", "bench.py"],
                            capture_output=True, text=True).stdout.strip())
d["_commit"] = head + ("+uncommitted" if dirty else "")
json.dump(d, open(dst, "w"), indent=1)
print(dst, d["_commit"])

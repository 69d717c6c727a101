import sys, json, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.atlas_mesh_extraction("cuda:0"), indent=1))

import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]; w=r["hbm_per_wave"]
print("nodes/ray %.2f tris/ray %.2f | per 64-ray batch: node records %.1f, tri records %.1f" % (r["nodes_per_ray"], r["tris_per_ray"], w["node_records_per_ray"]*64, w["tri_records_per_ray"]*64))

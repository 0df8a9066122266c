import sys, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import raytrace_amd as rt
r = rt.Renderer(0); r.upload(rt.random_scene(11)); cam = rt.camera_default(1920,1080)
job = rt.Job(1920,1080,64,seed=0,flags=rt.FLAG_PROFILE)
r.render_host(cam, job, debug=False)
r.counters(reset=True)
r.render_host(cam, job, debug=False)
c = r.counters()
t0=c["first_wave_start"]
print(json.dumps({"kernel_ms": r.trace_ms(0), "last_start_ms": (c["last_wave_start"]-t0)/1e5, "first_end_ms": (c["first_wave_end"]-t0)/1e5, "last_end_ms": (c["last_wave_end"]-t0)/1e5, "avg_life_ms": c["realtime_total"]/5120/1e5, "trips": c["wave_iterations"]}))

import ctypes as C, numpy as np
log = np.zeros(5120, np.uint64)
rt.N.check(rt.N.device.rt_debug_wave_log(r._h, C.c_void_p(log.ctypes.data), len(log)), "wave_log")
life = (log >> np.uint64(24)).astype(np.float64) / 1e5
trips = (log & np.uint64(0xffffff)).astype(np.int64)
q = [0, 1, 5, 25, 50, 75, 90, 95, 99, 99.9, 100]
print(json.dumps({"lifetime_ms_percentiles": dict(zip(map(str, q), np.round(np.percentile(life, q), 3).tolist())),
                  "trips_percentiles": dict(zip(map(str, q), np.percentile(trips, q).tolist())),
                  "us_per_trip_percentiles": dict(zip(map(str, q), np.round(np.percentile(life * 1e3 / np.maximum(trips, 1), q), 2).tolist()))}))
order = np.argsort(life)
print("longest-lived waves (life ms, trips):", [(round(float(life[i]), 3), int(trips[i])) for i in order[-8:]])
print("shortest-lived waves:", [(round(float(life[i]), 3), int(trips[i])) for i in order[:8]])

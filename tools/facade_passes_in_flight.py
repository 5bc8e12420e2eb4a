"""Frames per second of the kept C++ render loop (Scene / PathTracer::UpdateDeviceScene + Render once per frame) with one
pass at a time and with passes in flight.   tools/facade_passes_in_flight.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from nexus_amd import capi, pod
from tests import test_host_facade as T

W, H = 1920, 1080
for R in (1, 3, 6):
    sc = T._cornell_facade(W, H, 8)
    pt = capi.PathTracer(W, H)
    pt.set_modes(pod.RNG_PIXEL_KEYED, pod.COMPACT_FAST, pod.CONDUCTOR_EXTENDED)
    pt.update_device_scene(sc)
    pt.set_passes_in_flight(R)
    for _ in range(16):
        pt.update_device_scene(sc)
        pt.render(sc)
    pt.read_pixels()
    t0 = time.perf_counter()
    n = 96
    for _ in range(n):
        pt.update_device_scene(sc)   # what Renderer::Render does every frame (Renderer.cpp:41-77)
        pt.render(sc)
    pt.read_pixels()
    dt = time.perf_counter() - t0
    print("passes in flight %d: %.2f ms per frame, %.0f Msamples/s" % (R, dt / n * 1e3, W * H * n / dt / 1e6), flush=True)
    pt.close()

"""First end-to-end check on a GPU: HIP library against the oracle on the Cornell box."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oracle import formats
from path_tracer_amd import hip_api

D = 'tests/golden/data'
sc = formats.load_scene(f'{D}/cornell_box.vks', f'{D}/cornell_box_textures')
lights = formats.load_lights(f'{D}/cornell_box.lights')
spec = formats.load_rt_save(f'{D}/saves/cornell_box/default.rt_save')
print(hip_api.random_numbers(0, 0, 0, 2), oracle.rng_stream(0, 0, 0, 2)[0], flush=True)
textures = [(t.width, t.height, t.format, [t.payload[int(t.mip_offsets[m]):int(t.mip_offsets[m]) + t.mip_extents[m][2]] for m in range(t.mip_count)]) for t in sc.textures]
t0 = time.time()
gs = hip_api.Scene(len(sc.material_names), sc.dequantization_factor, sc.dequantization_summand, sc.positions, sc.normals_uvs, sc.materials, textures)
info = gs.bvh_info()
print('scene created in', time.time() - t0, 'nodes', info.node_count, 'tris', info.triangle_count, 'depth', info.depth, 'build ms', info.build_ms, flush=True)
gs.set_lights(lights)
os_ = oracle.OracleScene(sc)
rng = np.random.default_rng(1)
N = 100000
o = rng.uniform([0, -0.56, 0], [0.556, 0.4, 0.549], (N, 3)).astype(np.float32)
d = rng.normal(size=(N, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
a = os_.trace_rays(o, d); b = gs.trace_rays(o, d)
print('rays: tri mismatch', (a[0] != b[0]).sum(), 'tuv mismatch', (a[1].view(np.uint32) != b[1].view(np.uint32)).any(axis=1).sum(), 'front mismatch', (a[2] != b[2]).sum(), flush=True)
W = H = 128
c = oracle.write_constants(spec, W, H, sc, lights)
gc = hip_api.Constants.from_buffer_copy(bytes(c))
gs.create_render_targets(W, H)
pa = os_.primary_hits(c, 2696); pb = gs.trace_primary_rays(gc, 2696)
print('primary mismatch', (pa[0] != pb[0]).sum(), (pa[1].view(np.uint32) != pb[1].view(np.uint32)).sum(), (pa[2] != pb[2]).sum(), flush=True)
for strat in (3, 2, 1, 0):
    gs.set_render_settings(strat, 4, sc.emission_material_index)
    ia, ra = os_.render(c, lights, strategy=strat, path_length=4, first_frame_index=2696, frame_count=8, threads=16)
    gs.reset_statistics()
    gs.render_samples(gc, 2696, 8, 1, accumulate=False)
    ib, n = gs.read_accumulation()
    st = gs.statistics()
    neq = (ia.view(np.uint32) != ib.view(np.uint32)).any(axis=2).sum()
    print('strategy', strat, 'rays', ra, st.extension_rays + st.shadow_rays, 'spp', n, 'pixels differing', neq, 'max abs diff', np.nanmax(np.abs(ia - ib)), 'nan', np.isnan(ia).sum(), np.isnan(ib).sum(), flush=True)
# throughput at 1080p
W, H = 1920, 1080
c = oracle.write_constants(spec, W, H, sc, lights)
gc = hip_api.Constants.from_buffer_copy(bytes(c))
gs.create_render_targets(W, H)
gs.set_render_settings(3, 4, sc.emission_material_index)
gs.render_samples(gc, 2696, 4, 1, accumulate=False); gs.synchronize()
gs.reset_statistics(); gs.set_instrumentation(True, False)
t0 = time.time(); gs.render_samples(gc, 2700, 32, 1, accumulate=True); gs.synchronize(); dt = time.time() - t0
st = gs.statistics()
rays = st.extension_rays + st.shadow_rays
print('1080p 32 spp', dt, 's', rays / dt / 1e6, 'Mrays/s', W * H * 32 / dt / 1e6, 'Msamples/s', flush=True)
for i, name in enumerate(hip_api.KERNEL_NAMES):
    print(name, st.kernel_ms[i], 'ms', st.kernel_launches[i], 'launches')
img, n = gs.read_accumulation()
np.save('gpurun_out/cornell_1080p.npy', (img[::4, ::4, :3] / n).astype(np.float16))

"""Checks of the samplers, densities and the BRDF that do not depend on any implementation of this repository:
a probability density integrates to one, a sampler's directions are distributed like its density says (chi-square
on a histogram over the sphere), and the radiance a surface reflects under a uniform sky equals its directional
albedo, computed here from a restatement of the BRDF formula in double precision with numpy.

Every check takes a callable evaluate(function, shading_data, inputs) -> (n, 4) floats with the numbering
PT_FUNCTION_* of include/pt_hip.h, so the same checks run against the CPU oracle, against the kernel bodies
compiled for the host and against the HIP library on a GPU.

Shader lines cited: src/shaders/brdfs.glsl:19-46 (frostbite_brdf), :58-121 (GGX sampler and density), :139-153 (projected
solid angle), :174-208 (mixture); src/shaders/pathtrace.frag.glsl:50-64 (spherical), :69-175 (spherical lights).
"""
import numpy as np

(SAMPLE_GGX_IN_DIR, GGX_IN_DIR_DENSITY, SAMPLE_HEMISPHERE_PSA, HEMISPHERE_PSA_DENSITY, SAMPLE_HEMISPHERE_SPHERICAL,
 HEMISPHERE_SPHERICAL_DENSITY, SAMPLE_FROSTBITE_BRDF, FROSTBITE_BRDF_DENSITY, FROSTBITE_BRDF, SAMPLE_LIGHTS, LIGHTS_DENSITY) = range(11)


def frame_about(normal):
    """Two unit vectors that complete normal to a right-handed orthonormal basis (any such basis will do)."""
    normal = np.asarray(normal, np.float64)
    helper = np.array([1.0, 0.0, 0.0]) if abs(normal[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    tangent = np.cross(helper, normal)
    tangent /= np.linalg.norm(tangent)
    return tangent, np.cross(normal, tangent)


def make_shading_data(normal, out_dir, diffuse_albedo=(0.5, 0.4, 0.3), fresnel_0=(0.04, 0.04, 0.04), roughness=0.3, pos=(0.0, 0.0, 0.0)):
    """The 20 floats of shading_data_t (shading_data.glsl:19-39) for unit vectors normal and out_dir."""
    normal = np.asarray(normal, np.float64) / np.linalg.norm(normal)
    out_dir = np.asarray(out_dir, np.float64) / np.linalg.norm(out_dir)
    normal32, out32 = normal.astype(np.float32), out_dir.astype(np.float32)
    lambert_out = np.float32((normal32[0] * out32[0] + normal32[1] * out32[1]) + normal32[2] * out32[2])
    return np.concatenate([np.asarray(pos, np.float32), normal32, out32, [lambert_out], [0, 0, 0], np.asarray(diffuse_albedo, np.float32),
                           np.asarray(fresnel_0, np.float32), [roughness]]).astype(np.float32)


def directions_on_grid(normal, theta_edges, phi_edges, sub):
    """Midpoints of sub x sub cells inside every bin of a grid in the polar angle theta about normal and the azimuth
    phi. Returns directions (bins_theta, bins_phi, sub * sub, 3) and the solid angle sin(theta) d(theta) d(phi) of
    every sub-cell (bins_theta, bins_phi, sub * sub)."""
    tangent, bitangent = frame_about(normal)
    normal = np.asarray(normal, np.float64) / np.linalg.norm(normal)
    offsets = (np.arange(sub) + 0.5) / sub
    theta = theta_edges[:-1, None] + np.diff(theta_edges)[:, None] * offsets[None, :]
    phi = phi_edges[:-1, None] + np.diff(phi_edges)[:, None] * offsets[None, :]
    t = theta[:, None, :, None]
    p = phi[None, :, None, :]
    dirs = (np.sin(t) * np.cos(p))[..., None] * tangent + (np.sin(t) * np.sin(p))[..., None] * bitangent + (np.cos(t) + 0 * p)[..., None] * normal
    solid_angle = np.sin(t) * (np.diff(theta_edges)[:, None, None, None] / sub) * (np.diff(phi_edges)[None, :, None, None] / sub) + 0 * p
    shape = (len(theta_edges) - 1, len(phi_edges) - 1, sub * sub)
    return dirs.reshape(shape + (3,)), solid_angle.reshape(shape)


def integrate_density(evaluate, density_function, shading_data, normal, component=0, cells=(1024, 2048)):
    """The integral of a density over the sphere of directions by the midpoint rule on a grid that is uniform in the
    polar angle theta about normal and the azimuth (solid angle sin(theta) d(theta) d(phi), which also tames a
    density that grows like 1 / sin(theta) at the pole)."""
    tangent, bitangent = frame_about(normal)
    unit_normal = np.asarray(normal, np.float64) / np.linalg.norm(normal)
    theta = (np.arange(cells[0]) + 0.5) * (np.pi / cells[0])
    phi = -np.pi + (np.arange(cells[1]) + 0.5) * (2.0 * np.pi / cells[1])
    total = 0.0
    # in slabs, so that the arrays stay small
    for begin in range(0, cells[0], 64):
        t = theta[begin:begin + 64, None]
        dirs = (np.sin(t) * np.cos(phi))[..., None] * tangent + (np.sin(t) * np.sin(phi))[..., None] * bitangent + (np.cos(t) + 0 * phi)[..., None] * unit_normal
        values = evaluate(density_function, shading_data, dirs.reshape(-1, 3))[:, component].astype(np.float64).reshape(dirs.shape[:2])
        total += float((values * np.sin(t)).sum()) * (np.pi / cells[0]) * (2.0 * np.pi / cells[1])
    return total


def chi_square_of_sampler(evaluate, sample_function, density_function, shading_data, normal, component=0, sample_count=400000, bins=(24, 48), sub=12, seed=1):
    """Draws sample_count directions with the sampler, bins them over the sphere about normal and compares the counts
    with the integral of the density over each bin (bins uniform in theta and phi). Returns (chi square, degrees of freedom, total expected
    probability, share of samples that are unit vectors)."""
    rng = np.random.default_rng(seed)
    randoms = rng.random((sample_count, 2), dtype=np.float32)
    dirs = evaluate(sample_function, shading_data, randoms)[:, :3].astype(np.float64)
    lengths = np.linalg.norm(dirs, axis=1)
    unit = np.abs(lengths - 1.0) < 1e-3
    tangent, bitangent = frame_about(normal)
    unit_normal = np.asarray(normal, np.float64) / np.linalg.norm(normal)
    d = dirs[unit]
    theta = np.arccos(np.clip(d @ unit_normal, -1.0, 1.0))
    phi = np.arctan2(d @ bitangent, d @ tangent)
    theta_edges = np.linspace(0.0, np.pi, bins[0] + 1)
    phi_edges = np.linspace(-np.pi, np.pi, bins[1] + 1)
    observed, _, _ = np.histogram2d(theta, phi, bins=(theta_edges, phi_edges))
    grid_dirs, solid_angle = directions_on_grid(normal, theta_edges, phi_edges, sub)
    density = evaluate(density_function, shading_data, grid_dirs.reshape(-1, 3))[:, component].astype(np.float64).reshape(solid_angle.shape)
    probability = (density * solid_angle).sum(axis=2)
    expected = probability * len(d)
    # Bins that expect few samples are pooled into one
    large = expected >= 20.0
    observed_pooled = np.append(observed[large], observed[~large].sum())
    expected_pooled = np.append(expected[large], expected[~large].sum())
    keep = expected_pooled > 0
    # Samples where the density is zero all over a bin: a few on the silhouette of a light, where the bin's sub-cells
    # all lie outside, are rounding; more than that is a sampler that leaves the support of its density
    if observed_pooled[~keep].sum() > 2e-4 * len(d):
        return np.inf, int(keep.sum()), float(probability.sum()), float(unit.mean())
    chi_square = float((((observed_pooled - expected_pooled) ** 2)[keep] / expected_pooled[keep]).sum())
    return chi_square, int(keep.sum()) - 1, float(probability.sum()), float(unit.mean())


def chi_square_limit(degrees_of_freedom):
    """A chi-square value this large or larger has a probability below 1e-9 for a correct sampler (Wilson-Hilferty
    approximation at six standard deviations); errors of normalisation or of a Jacobian exceed it by orders of
    magnitude at the sample counts used here."""
    k = float(degrees_of_freedom)
    return k * (1.0 - 2.0 / (9.0 * k) + 6.0 * np.sqrt(2.0 / (9.0 * k))) ** 3


# ---- The BRDF in double precision, restated from src/shaders/brdfs.glsl:7-46 ----

def fresnel_schlick(f_0, f_90, lambert):
    flip = 1.0 - lambert
    return flip ** 5 * (f_90 - f_0) + f_0


def frostbite_brdf(normal, out_dir, in_dirs, diffuse_albedo, fresnel_0, roughness):
    """frostbite_brdf() for directions in_dirs (n, 3): (n, 3) RGB values."""
    normal, out_dir = (np.asarray(v, np.float64) for v in (normal, out_dir))
    diffuse_albedo, fresnel_0 = (np.asarray(v, np.float64) for v in (diffuse_albedo, fresnel_0))
    lambert_out = normal @ out_dir
    lambert_in = in_dirs @ normal
    half_dir = in_dirs + out_dir
    half_dir /= np.linalg.norm(half_dir, axis=1, keepdims=True)
    half_dot_out = half_dir @ out_dir
    f_90 = half_dot_out ** 2 * (2.0 * roughness) + 0.5
    fresnel_diffuse = fresnel_schlick(1.0, f_90, lambert_out) * fresnel_schlick(1.0, f_90, lambert_in)
    brdf = fresnel_diffuse[:, None] * diffuse_albedo
    half_dot_normal = half_dir @ normal
    roughness_2 = roughness * roughness
    ggx = (roughness_2 * half_dot_normal - half_dot_normal) * half_dot_normal + 1.0
    ggx = roughness_2 / (ggx * ggx)
    masking = lambert_in * np.sqrt((lambert_out - roughness_2 * lambert_out) * lambert_out + roughness_2)
    shadowing = lambert_out * np.sqrt((lambert_in - roughness_2 * lambert_in) * lambert_in + roughness_2)
    smith = 0.5 / (masking + shadowing)
    fresnel = fresnel_schlick(fresnel_0[None, :], 1.0, np.maximum(0.0, half_dot_out)[:, None])
    brdf = (brdf + (ggx * smith)[:, None] * fresnel) / np.pi
    brdf[np.minimum(lambert_in, lambert_out) < 0.0] = 0.0
    return brdf


def directional_albedo(normal, out_dir, diffuse_albedo, fresnel_0, roughness, cells=(1024, 2048)):
    """The integral of frostbite_brdf() times the cosine over the hemisphere of incoming directions: what a surface
    reflects towards out_dir under a sky of unit radiance."""
    theta_edges = np.linspace(0.0, 0.5 * np.pi, cells[0] + 1)
    phi_edges = np.linspace(-np.pi, np.pi, cells[1] + 1)
    total = np.zeros(3)
    unit_normal = np.asarray(normal, np.float64) / np.linalg.norm(normal)
    for begin in range(0, cells[0], 64):
        dirs, solid_angle = directions_on_grid(normal, theta_edges[begin:begin + 65], phi_edges, 1)
        flat = dirs.reshape(-1, 3)
        values = frostbite_brdf(unit_normal, out_dir, flat, diffuse_albedo, fresnel_0, roughness) * (flat @ unit_normal)[:, None]
        total += (values * solid_angle.reshape(-1, 1)).sum(axis=0)
    return total

"""Synthetic scenes of BASELINE.json's configs, built through the `su_*` protocol.

The reference ships no scenes (its data/ directory is git-ignored), so these are
generated: a Cornell box out of `Rectangle` props (reference built-in shape id 5,
src/core/resource/manager.zig:36-44), two emissive quads as unoccluding `Light` entities
(util/scene_loader.zig:249,365-366), the classic short/tall boxes as triangle meshes, and a
procedural ~1 M-triangle mesh: a closed displaced sphere hanging in the box, and a displaced
heightfield lying on its floor as a second, lighter variant.
"""
import json

import numpy as np

RECTANGLE = 5

WHITE = [0.725, 0.71, 0.68]
RED = [0.63, 0.065, 0.05]
GREEN = [0.14, 0.45, 0.091]
MESH_CENTRE = (0.0, 1.2, 0.0)  # the 1 M-triangle mesh centred in the box at ~60 % of its height (the box is 2 high)


def trafo(r0, r1, scale, position):
    """Row-major 4x4 as su_prop_set_transformation expects: rows = scaled basis vectors, row 3 = translation
    (reference src/capi/capi.zig:491-499).  r2 = r0 x r1 is the Rectangle's normal (rectangle.zig:30-32)."""
    r0 = np.asarray(r0, dtype=np.float64)
    r1 = np.asarray(r1, dtype=np.float64)
    r2 = np.cross(r0, r1)
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, :3] = r0 * scale[0]
    m[1, :3] = r1 * scale[1]
    m[2, :3] = r2 * scale[2]
    m[3, :3] = position
    m[3, 3] = 1.0
    return m


def lambert(color):
    # Debug-style Lambert with a per-material albedo (debug_sample.zig:19-47)
    return json.dumps({"rendering": {"Debug": {"color": color}}})


def substitute(color, roughness=0.8, metallic=0.0):
    return json.dumps({"rendering": {"Substitute": {"color": color, "roughness": roughness, "metallic": metallic}}})


def substitute_textured(image_id, roughness=0.8, metallic=0.0, sampler=None, scale=None):
    """Substitute with an image colour map (material_provider.zig:438-632: {"usage": "Color", "id": n, ...})."""
    color = {"usage": "Color", "id": int(image_id)}
    if sampler:
        color["sampler"] = sampler
    if scale is not None:
        color["scale"] = scale
    return json.dumps({"rendering": {"Substitute": {"color": color, "roughness": roughness, "metallic": metallic}}})


def substitute_full(color=(0.5, 0.5, 0.5), **body):
    """Substitute with any of the reference's parameters (material_provider.zig:262-342): roughness, metallic, specular, ior,
    anisotropy, anisotropy_rotation, translucency, normal, coating {color, attenuation_distance, ior, normal, roughness, scale,
    thickness}, flakes {color, coverage, roughness, size}, two_sided ...; a texture is tex(image_id, usage)."""
    b = {"color": color if isinstance(color, dict) else list(color)}
    b.update(body)
    return json.dumps({"rendering": {"Substitute": b}})


def tex(image_id, usage="Weight", sampler=None, scale=None):
    """A texture description (material_provider.zig:438-632)."""
    t = {"usage": usage, "id": int(image_id)}
    if sampler:
        t["sampler"] = sampler
    if scale is not None:
        t["scale"] = scale
    return t


def weight_image(n=8, lo=0.1, hi=0.9, dtype=np.float32):
    """n x n one-channel image with distinct texels in [lo, hi] (roughness / metallic / coating scale ... maps)."""
    y, x = np.mgrid[0:n, 0:n]
    w = lo + (hi - lo) * (((x * 5 + y * 3) % n) / (n - 1.0))
    return (w * 255.0 + 0.5).astype(np.uint8) if dtype == np.uint8 else w.astype(np.float32)


def normal_image(n=16, strength=0.5, dtype=np.uint8):
    """n x n two-channel tangent-space normal map (xy of a bumpy normal; z is reconstructed, material_helper.zig:18-20):
    uint8 = snorm (byte / 128 - 1), float32 as is."""
    y, x = np.mgrid[0:n, 0:n]
    nx = strength * np.sin(2.0 * np.pi * (x + 0.5) / n * 2.0)
    ny = strength * np.cos(2.0 * np.pi * (y + 0.5) / n * 3.0)
    xy = np.stack([nx, ny], axis=-1)
    if dtype == np.uint8:
        return np.clip(np.floor((xy + 1.0) * np.where(xy > 0.0, 127.5, 128.0)), 0, 255).astype(np.uint8)  # encoding.zig:14-16
    return xy.astype(np.float32)


def checker_image(n=8, dtype=np.float32):
    """n x n test image with distinct texels (a smooth ramp times a checker), float32 linear or uint8 sRGB."""
    y, x = np.mgrid[0:n, 0:n]
    ramp = np.stack([(x + 1) / n, (y + 1) / n, 1.0 - (x + y) / (2.0 * n)], axis=-1)
    img = ramp * np.where((x + y) % 2 == 0, 0.85, 0.35)[..., None]
    return (img * 255.0 + 0.5).astype(np.uint8) if dtype == np.uint8 else img.astype(np.float32)


def mask_image(n=16, dtype=np.uint8):
    """n x n one-channel opacity mask: holes (0), solid parts (1) and a band of partial opacities, so that both draws of
    stochasticOpacity (material_base.zig:44-55) happen."""
    y, x = np.mgrid[0:n, 0:n]
    r = np.hypot(x - (n - 1) / 2.0, y - (n - 1) / 2.0) / (n / 2.0)
    o = np.where((x // 2 + y // 2) % 3 == 0, 0.0, 1.0)
    o = np.where((r > 0.45) & (r < 0.8), 0.25 + 0.5 * ((x * 7 + y * 3) % 8) / 8.0, o)
    return (o * 255.0 + 0.5).astype(np.uint8) if dtype == np.uint8 else o.astype(np.float32)


def with_mask(material_json, image_id, sampler=None, scale=None):
    """Adds "mask" (material_provider.zig:177, 244, 265: usage Opacity) to a Glass / Light / Substitute material JSON."""
    m = json.loads(material_json)
    (body,) = m["rendering"].values()
    body["mask"] = {"usage": "Opacity", "id": int(image_id)}
    if sampler:
        body["mask"]["sampler"] = sampler
    if scale is not None:
        body["mask"]["scale"] = scale
    return json.dumps(m)


def glass(ior=1.5, color=(1.0, 1.0, 1.0), distance=1.0, priority=0, **more):
    """Thick specular dielectric (glass_material.zig); `color` is reached after `distance` inside the medium."""
    body = {"ior": ior, "attenuation_color": list(color), "attenuation_distance": distance, "priority": priority}
    body.update(more)
    return json.dumps({"rendering": {"Glass": body}})


def light_material(spectrum=(17.0, 12.0, 4.0), value=1.0, num_samples=1, emission_map=None, sampler=None):
    """Light material; emission_map: an image id -> the emittance is multiplied by the map (emittance.zig:49) and the light
    is importance-sampled by it (light_material.zig:52-125)."""
    e = {"spectrum": list(spectrum), "value": value, "num_samples": num_samples}
    if emission_map is not None:
        e["emission_map"] = {"usage": "Emission", "id": int(emission_map)}
        if sampler:
            e["emission_map"]["sampler"] = sampler
    return json.dumps({"rendering": {"Light": {"emittance": e}}})


def sky_image(width=64, height=32, sun=(0.62, 0.3), uniform=None):
    """A small float32 lat-long / disc environment map: a vertical gradient plus a bright blob ("sun") around uv = sun;
    uniform: a constant value instead."""
    if uniform is not None:
        return np.full((height, width, 3), uniform, dtype=np.float32)
    v, u = np.mgrid[0:height, 0:width]
    u = (u + 0.5) / width
    v = (v + 0.5) / height
    base = np.stack([0.25 + 0.3 * v, 0.4 + 0.25 * v, 0.9 - 0.5 * v], axis=-1)
    d2 = (u - sun[0]) ** 2 + ((v - sun[1]) * height / width) ** 2
    blob = 40.0 * np.exp(-d2 / 0.002)
    return (base + blob[..., None] * np.array([1.0, 0.9, 0.7])).astype(np.float32)


def box_mesh(lo, hi, rot_y_deg=0.0, subdivide=1, first_face=None):
    """Axis-aligned box [lo, hi] rotated about the y axis through its centre, face normals pointing outward.  Every face is a
    subdivide x subdivide grid of quads, two triangles each (subdivide = 1: 24 vertices, 12 triangles; 3: 108; 9: 972);
    first_face = (nu, nv): the first face's grid instead (e.g. (1, 4): 8 triangles there, 18 in all)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    c = 0.5 * (lo + hi)
    h = 0.5 * (hi - lo)
    faces = [  # (normal, u axis, v axis) with u x v = normal
        ((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),
        ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, -1, 0), (1, 0, 0), (0, 0, 1)),
        ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0)),
    ]
    a = np.radians(rot_y_deg)
    rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    pos, nrm, uv, idx = [], [], [], []
    for f, (n, u, v) in enumerate(faces):
        n, u, v = (np.asarray(x, dtype=np.float64) for x in (n, u, v))
        nu, nv = first_face if (first_face and 0 == f) else (subdivide, subdivide)
        base = len(pos)
        for j in range(nv + 1):
            for i in range(nu + 1):
                su, sv = 2.0 * i / nu - 1.0, 2.0 * j / nv - 1.0
                p = (n + su * u + sv * v) * h
                pos.append(rot @ p + c)
                nrm.append(rot @ n)
                uv.append((0.5 * (su + 1), 0.5 * (sv + 1)))
        for j in range(nv):
            for i in range(nu):
                a0 = base + j * (nu + 1) + i
                a1, a2, a3 = a0 + 1, a0 + nu + 2, a0 + nu + 1
                idx += [a0, a1, a2, a0, a2, a3]
    return (np.asarray(idx, dtype=np.uint32), np.asarray(pos, dtype=np.float32), np.asarray(nrm, dtype=np.float32),
            np.asarray(uv, dtype=np.float32))


def quad_mesh(n=1, size=1.0):
    """The Rectangle shape (xy in [-size/2, size/2], normal +z, rectangle.zig:30-62) as an n x n grid of quads, two
    triangles each, wound so that cross(e1, e2) points along +z like the Rectangle's normal."""
    xs = np.linspace(-0.5 * size, 0.5 * size, n + 1, dtype=np.float32)
    gx, gy = np.meshgrid(xs, xs, indexing="xy")
    pos = np.stack([gx.ravel(), gy.ravel(), np.zeros((n + 1) * (n + 1), dtype=np.float32)], axis=1).astype(np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (pos.shape[0], 1))
    uv = (pos[:, :2] / size + 0.5).astype(np.float32)
    tris = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            b, c, d = a + 1, a + n + 2, a + n + 1
            tris += [(a, b, c), (a, c, d)]
    return np.array(tris, dtype=np.uint32), pos, nrm, uv


def _value_noise_lattice(n, seed_seq):
    """n x n lattice of PCG32 floats (seed 0, sequence seed_seq), pure numpy-free loop kept tiny."""
    state = 0
    inc = ((seed_seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
    mul = 6364136223846793005
    mask = 0xFFFFFFFFFFFFFFFF

    def step():
        nonlocal state
        old = state
        state = (old * mul + inc) & mask
        xrs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xrs >> rot) | (xrs << ((-rot) & 31))) & 0xFFFFFFFF

    step()
    step()
    out = np.empty(n * n, dtype=np.float32)
    for i in range(n * n):
        out[i] = step() / 4294967296.0
    return out.reshape(n, n)


def heightfield_mesh(n=708, size=1.2, height=0.35, octaves=5):
    """n x n vertices -> 2 (n-1)^2 triangles (n = 708: 999 698), 5-octave value noise displacement over
    a square of edge `size` centred at the origin in the xz plane; smooth normals; uvs."""
    u = np.linspace(0.0, 1.0, n, dtype=np.float64)
    uu, vv = np.meshgrid(u, u, indexing="xy")
    hgt = np.zeros((n, n), dtype=np.float64)
    amp = 1.0
    total = 0.0
    for o in range(octaves):
        cells = 4 << o
        lat = _value_noise_lattice(cells + 1, o + 1).astype(np.float64)
        x = uu * cells
        y = vv * cells
        x0 = np.minimum(x.astype(np.int64), cells - 1)
        y0 = np.minimum(y.astype(np.int64), cells - 1)
        fx = x - x0
        fy = y - y0
        fx = fx * fx * (3 - 2 * fx)
        fy = fy * fy * (3 - 2 * fy)
        v00 = lat[y0, x0]
        v10 = lat[y0, x0 + 1]
        v01 = lat[y0 + 1, x0]
        v11 = lat[y0 + 1, x0 + 1]
        hgt += amp * ((v00 * (1 - fx) + v10 * fx) * (1 - fy) + (v01 * (1 - fx) + v11 * fx) * fy)
        total += amp
        amp *= 0.5
    hgt = hgt / total
    # dome-like envelope so the mesh sits inside the box
    env = np.sin(np.pi * uu) * np.sin(np.pi * vv)
    y = height * (0.25 + 0.75 * hgt) * env
    px = (uu - 0.5) * size
    pz = (vv - 0.5) * size
    pos = np.stack([px, y, pz], axis=-1)

    dx = np.gradient(y, axis=1) / (size / (n - 1))
    dz = np.gradient(y, axis=0) / (size / (n - 1))
    nrm = np.stack([-dx, np.ones_like(dx), -dz], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)

    ii, jj = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="xy")
    v0 = (jj * n + ii).reshape(-1)
    v1 = v0 + 1
    v2 = v0 + n
    v3 = v2 + 1
    # counter-clockwise seen from +y
    idx = np.stack([v0, v2, v1, v1, v2, v3], axis=-1).reshape(-1).astype(np.uint32)
    uv = np.stack([uu, vv], axis=-1)
    return (idx, pos.reshape(-1, 3).astype(np.float32), nrm.reshape(-1, 3).astype(np.float32),
            uv.reshape(-1, 2).astype(np.float32))


def _pcg32_floats(count, seed_seq):
    """`count` PCG32 floats of the stream (seed 0, sequence seed_seq), vectorised: the LCG state after k steps is
    a_k * s0 + c_k with (a_k, c_k) built by doubling, so the whole run is a handful of uint64 array operations."""
    mul = np.uint64(6364136223846793005)
    inc = np.uint64(((seed_seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        a = np.ones(1, dtype=np.uint64)
        c = np.zeros(1, dtype=np.uint64)
        while a.size < count + 2:  # (a, c)[k] for k = 0 .. size-1; doubling: step size more from every entry
            am, cm = a[-1] * mul, c[-1] * mul + inc  # the map of `size` steps
            a, c = np.concatenate([a, a * am]), np.concatenate([c, c * am + cm])
        # as _value_noise_lattice: state 0, two steps, then every output is computed from the state *before* its step
        s0 = np.uint64(0)
        old = (a * s0 + c)[2:count + 2]
        xrs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        out = (xrs >> rot) | (xrs << ((np.uint32(32) - rot) & np.uint32(31)))
    return (out.astype(np.float64) / 4294967296.0).astype(np.float32)


def displaced_sphere_mesh(n=708, radius=0.52, amplitude=0.22, octaves=5):
    """a closed displaced sphere, n x n vertices -> 2 (n-1)^2 triangles (n = 708: 999 698), 5-octave value
    noise displacement (3-D lattices of PCG32 floats, seed 0, sequences 1..octaves), smooth normals, uvs, centred at the origin.
    The vertex grid is the octahedral map of [-1, 1]^2 onto the sphere; the square's border folds onto itself (the vertices
    (u, 1) and (-u, 1) are the same point), so the surface is closed; the twins share their normals."""
    assert 0 == n % 2
    half = -1.0 + 2.0 * np.arange(n // 2, dtype=np.float64) / (n - 1)
    u = np.concatenate([half, -half[::-1]])  # exactly symmetric: u[n-1-i] = -u[i]
    uu, vv = np.meshgrid(u, u, indexing="xy")
    z = 1.0 - np.abs(uu) - np.abs(vv)
    lower = z < 0.0
    x = np.where(lower, (1.0 - np.abs(vv)) * np.sign(uu), uu)
    y = np.where(lower, (1.0 - np.abs(uu)) * np.sign(vv), vv)
    d = np.stack([x, z, y], axis=-1)  # the octahedron's axis along +y: the poles are the top and the bottom of the mesh
    d /= np.linalg.norm(d, axis=-1, keepdims=True)

    hgt = np.zeros((n, n), dtype=np.float64)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        cells = 4 << o
        m = cells + 1
        lat = _pcg32_floats(m * m * m, o + 1).astype(np.float64).reshape(m, m, m)
        q = (d * 0.5 + 0.5) * cells
        q0 = np.minimum(q.astype(np.int64), cells - 1)
        f = q - q0
        f = f * f * (3 - 2 * f)
        acc = 0.0
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    wgt = (f[..., 0] if dx else 1 - f[..., 0]) * (f[..., 1] if dy else 1 - f[..., 1]) * (f[..., 2] if dz else 1 - f[..., 2])
                    acc = acc + wgt * lat[q0[..., 2] + dz, q0[..., 1] + dy, q0[..., 0] + dx]
        hgt += amp * acc
        total += amp
        amp *= 0.5
    hgt /= total
    r = radius * (1.0 + amplitude * (2.0 * hgt - 1.0))
    pos = d * r[..., None]

    ii, jj = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="xy")
    v0 = (jj * n + ii).reshape(-1)
    v1, v2 = v0 + 1, v0 + n
    v3 = v2 + 1
    # (u, v) -> (x, z): counter-clockwise seen from outside is clockwise in the (u, v) square with y = up taken from the z of the map
    tri = np.stack([v0, v2, v1, v1, v2, v3], axis=-1).reshape(-1, 3)
    p = pos.reshape(-1, 3)
    fn = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    if np.sum(fn * p[tri[:, 0]]) < 0.0:  # outward winding, whatever the handedness of the axes above
        tri = tri[:, [0, 2, 1]]
        fn = -fn
    nrm = np.zeros_like(p)
    for k in range(3):
        np.add.at(nrm, tri[:, k], fn)
    nrm = nrm.reshape(n, n, 3)
    # the folded border: a vertex and its twin are one point of the surface
    for edge in (0, n - 1):
        row = nrm[edge] + nrm[edge, ::-1]
        nrm[edge] = row
        col = nrm[:, edge] + nrm[::-1, edge]
        nrm[:, edge] = col
    corner = nrm[0, 0] + nrm[0, n - 1] + nrm[n - 1, 0] + nrm[n - 1, n - 1]
    nrm[0, 0] = nrm[0, n - 1] = nrm[n - 1, 0] = nrm[n - 1, n - 1] = corner
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    uv = np.stack([uu * 0.5 + 0.5, vv * 0.5 + 0.5], axis=-1)
    return (tri.reshape(-1).astype(np.uint32), p.astype(np.float32), nrm.reshape(-1, 3).astype(np.float32),
            uv.reshape(-1, 2).astype(np.float32))


def cornell(engine, width, height, spp, integrator="PTMIS", material="lambert", boxes=True, mesh=None,
            max_depth=16, filter_name=None, split_threshold=None, light_samples=1, unoccluding_lights=True,
            mesh_material=None, mesh_lights=False, light_grid=1, alpha_transparency=False, camera_z=-3.9, many_lights=0,
            floor_material=None, spheres=False, sphere_lights=False, glass_objects=None, disk_and_cube=False, light_map=None,
            light_spectrum=(17.0, 12.0, 4.0), masked=None, disk_lights=False, two_sided_lights=False, clamp=None, lens=None,
            overrides=None, regularize=None, mesh_position=(0, 0.02, 0.1), box_subdivide=1, box_first_face=None, extra_props=0, portal_map=None,
            box_lift=0.0):
    """Builds the Cornell box of BASELINE.json's configs 1-3 on `engine` (zyg_amd.Engine).

    material: "lambert" (Debug-style Lambert) or "substitute"; mesh: optional (idx, pos, nrm, uv) placed at mesh_position (the
    default puts heightfield_mesh on the floor; displaced_sphere_mesh hangs at MESH_CENTRE, 60 % of the box's height).
    mesh_lights: the two emissive quads are triangle meshes (2 * light_grid^2 triangles each) instead of Rectangles.
    many_lights: n > 0 replaces the two quads by an n x n grid of small Rectangle lights under the ceiling (light-tree depth).
    spheres: two analytic Spheres (one Substitute-ish via `material`, one red) instead of / next to the boxes;
    sphere_lights: the two emitters are Spheres of diameter 0.3 instead of quads.
    glass_objects: None, or a dict: the short box turns into glass (`box` = a glass(...) JSON), a glass Sphere (`sphere`)
    stands in front of the tall box, optionally with a second dielectric inside it (`inner`, nested dielectrics);
    `outside`: a glass Sphere next to the box, in front of the void (alpha through glass with a Transparent sensor).
    disk_lights: the two emitters are Disks (diameter 0.5, the second one tilted) instead of quads; two_sided_lights: the
    lamp material is two-sided.
    masked: None, or a dict: `image` (a one-channel numpy mask) and optionally `sampler`, `scale`, `lights` (the lamps are masked
    too), `pane` (a masked thin-glass Rectangle): a masked Rectangle, Sphere and Disk stand in the box and the tall box's
    material is masked (opacity masks need Substitute / Light / Glass materials).
    clamp: the sensor's per-channel clamp, e.g. {"direct": 4, "indirect": 2} (sensor.zig:177-181, take_loader.zig "clamp").
    lens: thin-lens camera parameters, e.g. {"lens": {"radius": 0.05}, "focus": {"distance": 3.9}}
    (camera_perspective.zig:124-150, 260-270, aperture.zig:46-53).
    overrides: material JSONs by place: "floor", "back", "left", "right", "ceiling", "tall", "short", "sphere_a", "sphere_b".
    portal_map: a lat-long environment image: instead of the ceiling and the two lamps the box is lit through a Portal in the
    ceiling's place, a window onto a Dome with that map (scene/shape/portal.zig; the Dome itself becomes a hidden prototype).
    box_subdivide / box_first_face: the two boxes' triangle counts (box_mesh); extra_props: that many small Rectangle plaques on
    the back wall (the prop count: the device's prop tables hold 16).
    box_lift: the two boxes stand that far above the floor instead of on it (0: their bottom faces lie in the floor's plane, and
    which of the two coincident surfaces a ray meets first is decided by the last bit of two different formulas).
    Returns a dict with prop ids.
    """
    overrides = overrides or {}

    def pick(place, default):
        return engine.material(overrides[place]) if place in overrides else default

    mk = lambert if material == "lambert" else substitute
    cam = engine.camera(width, height, fov_deg=39.3)
    if lens:
        engine.camera_parameters(json.dumps(lens))
    engine.samples_per_pixel(spp)
    node = {"depth": {"surface": max_depth}}
    if split_threshold is not None:
        node["light_sampling"] = {"split_threshold": split_threshold}
    if regularize is not None:  # PathtracerMIS' roughness regularisation (take.zig:211, renderstate.zig:60-69)
        node["regularize_roughness"] = regularize
    engine.integrators(json.dumps({"surface": {integrator: node}}))
    sensor = {}
    if alpha_transparency:
        sensor["alpha_transparency"] = True
    if filter_name:
        sensor["filter"] = {filter_name: {}}
    if clamp:
        sensor["clamp"] = clamp
    engine.sensor(json.dumps(sensor))

    white = engine.material(mk(WHITE))
    red = engine.material(mk(RED))
    green = engine.material(mk(GREEN))
    # light_map: a numpy image -> the quad lights are textured (emission map x spectrum) and importance-sampled by it
    light_map_id = None if light_map is None else engine.image(light_map)
    lamp = engine.material(light_material(spectrum=light_spectrum, num_samples=light_samples, emission_map=light_map_id))

    mask_id = None
    if masked:
        mask_id = engine.image(masked["image"])
        mask_kw = {"sampler": masked.get("sampler"), "scale": masked.get("scale")}
        if masked.get("lights"):
            lamp = engine.material(with_mask(light_material(spectrum=light_spectrum, num_samples=light_samples), mask_id, **mask_kw))

    if two_sided_lights:
        lm = json.loads(light_material(spectrum=light_spectrum, num_samples=light_samples, emission_map=light_map_id))
        lm["rendering"]["Light"]["two_sided"] = True
        lamp = engine.material(json.dumps(lm))
    ids = {"camera": cam, "walls": [], "lights": [], "meshes": []}

    def wall(mat, r0, r1, scale, position):
        p = engine.prop(RECTANGLE, [mat])
        engine.transformation(p, trafo(r0, r1, scale, position))
        ids["walls"].append(p)
        return p

    # box: x in [-1, 1], y in [0, 2], z in [-1, 1]; all normals point inward; open towards -z (camera side)
    floor = pick("floor", white) if floor_material is None else engine.material(floor_material)
    wall(floor, (1, 0, 0), (0, 0, -1), (2, 2, 1), (0, 0, 0))   # floor, n = +y
    if portal_map is None:
        wall(pick("ceiling", white), (1, 0, 0), (0, 0, 1), (2, 2, 1), (0, 2, 0))    # ceiling, n = -y
    wall(pick("back", white), (-1, 0, 0), (0, 1, 0), (2, 2, 1), (0, 1, 1))   # back, n = -z
    wall(pick("left", red), (0, 0, -1), (0, 1, 0), (2, 2, 1), (-1, 1, 0))    # left, n = +x
    wall(pick("right", green), (0, 0, 1), (0, 1, 0), (2, 2, 1), (1, 1, 0))    # right, n = -x

    if many_lights:
        n = many_lights
        side = 1.6 / n
        for j in range(n):
            for i in range(n):
                p = engine.prop(RECTANGLE, [lamp], unoccluding=unoccluding_lights)
                cx, cz = -0.8 + (i + 0.5) * side, -0.8 + (j + 0.5) * side
                engine.transformation(p, trafo((1, 0, 0), (0, 0, 1), (0.6 * side, 0.5 * side, 1), (cx, 1.98 - 0.01 * ((i + j) % 3), cz)))
                engine.light(p)
                ids["lights"].append(p)
    if spheres:
        for pos_, d_, mat_ in (((-0.45, 0.4, 0.2), 0.8, pick("sphere_a", white)), ((0.5, 0.25, -0.3), 0.5, pick("sphere_b", red))):
            p = engine.prop(SPHERE, [mat_])
            engine.transformation(p, trafo((1, 0, 0), (0, 1, 0), (d_, d_, d_), pos_))
            ids["meshes"].append(p)
    if portal_map is not None:
        env = engine.material(light_material(spectrum=(1.0, 1.0, 1.0), emission_map=engine.image(portal_map), num_samples=light_samples))
        dome = engine.prop(DOME, [env])
        engine.transformation(dome, trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 0)))
        engine.light(dome)  # light 0: what the portal is a window onto
        window = engine.portal(0)
        engine.transformation(window, trafo((1, 0, 0), (0, 0, 1), (1.6, 1.6, 1), (0, 2.0, 0)))  # in the ceiling's plane, facing down
        engine.portal_light(window, 0)
        ids["lights"].append(window)
        ids["dome"] = dome
    for lx in (() if (many_lights or portal_map is not None) else (-0.45, 0.45)):  # two emissive quads just below the ceiling, facing down
        if disk_lights:  # Disk = shape 2; the second one tilted and lower, so that points see it at grazing angles
            p = engine.prop(2, [lamp], unoccluding=unoccluding_lights)
            if lx < 0:
                engine.transformation(p, trafo((1, 0, 0), (0, 0, 1), (0.5, 0.5, 1), (lx, 1.98, 0.0)))
            else:
                engine.transformation(p, trafo((0.8, 0.6, 0), (0, 0, 1), (0.6, 0.6, 1), (lx, 1.6, 0.1)))
            engine.light(p)
            ids["lights"].append(p)
            continue
        if sphere_lights:
            p = engine.prop(SPHERE, [lamp], unoccluding=unoccluding_lights)
            engine.transformation(p, trafo((1, 0, 0), (0, 1, 0), (0.3, 0.3, 0.3), (lx, 1.7, 0.0)))
            engine.light(p)
            ids["lights"].append(p)
            continue
        if mesh_lights:
            # the size is baked into the vertices: the reference's mesh-light areas and solid angles only hold for
            # uniform scales (triangle_mesh.zig:289-292 "HACK", 524-531)
            idx, pos, nrm, uv = quad_mesh(light_grid, 0.5)
            p = engine.prop(engine.triangle_mesh(idx, pos, nrm, uv), [lamp], unoccluding=unoccluding_lights)
            engine.transformation(p, trafo((1, 0, 0), (0, 0, 1), (1, 1, 1), (lx, 1.98, 0.0)))
        else:
            p = engine.prop(RECTANGLE, [lamp], unoccluding=unoccluding_lights)
            engine.transformation(p, trafo((1, 0, 0), (0, 0, 1), (0.5, 0.5, 1), (lx, 1.98, 0.0)))
        engine.light(p)
        ids["lights"].append(p)

    if disk_and_cube:  # a tilted Disk (shape 2) and a rotated, non-uniformly scaled Cube (shape 1)
        p = engine.prop(2, [red])
        engine.transformation(p, trafo((0.8, 0, -0.6), (0, 1, 0), (0.7, 0.7, 0.7), (-0.4, 0.45, -0.55)))
        ids["meshes"].append(p)
        c, s_ = np.cos(np.radians(25.0)), np.sin(np.radians(25.0))
        p = engine.prop(1, [green])
        engine.transformation(p, trafo((c, 0, -s_), (0, 1, 0), (0.5, 0.3, 0.4), (0.45, 0.15, -0.45)))
        ids["meshes"].append(p)
    if masked:
        cut_red = engine.material(with_mask(substitute(RED, 0.6), mask_id, **mask_kw))
        cut_green = engine.material(with_mask(substitute(GREEN, 0.3, 1.0), mask_id, **mask_kw))
        p = engine.prop(RECTANGLE, [cut_red])  # a screen facing the camera, in front of the short box
        engine.transformation(p, trafo((-1, 0, 0), (0, 1, 0), (0.9, 0.9, 1), (0.45, 0.8, -0.8)))
        ids["meshes"].append(p)
        p = engine.prop(SPHERE, [cut_green])
        engine.transformation(p, trafo((0.8, 0, 0.6), (0, 1, 0), (0.6, 0.6, 0.6), (-0.5, 1.55, -0.3)))
        ids["meshes"].append(p)
        p = engine.prop(2, [cut_red])  # Disk, lying above the floor: its shadow shows the mask
        engine.transformation(p, trafo((1, 0, 0), (0, 0.6, -0.8), (0.8, 0.8, 0.8), (-0.45, 0.5, -0.6)))
        ids["meshes"].append(p)
        if masked.get("pane"):
            pane = engine.material(with_mask(glass(1.5, (0.6, 0.9, 0.7), 0.05, thickness=0.02), mask_id, **mask_kw))
            p = engine.prop(RECTANGLE, [pane])
            engine.transformation(p, trafo((-1, 0, 0), (0, 0.96, 0.28), (1.2, 0.8, 1), (0.0, 1.3, 0.2)))
            ids["meshes"].append(p)
    identity = trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 0))
    glass_objects = glass_objects or {}
    for key, pos_, d_ in (("sphere", (-0.35, 0.36, -0.5), 0.7), ("inner", (-0.35, 0.36, -0.5), 0.4), ("outside", (1.7, 1.0, -1.2), 0.9)):
        if key in glass_objects:
            p = engine.prop(SPHERE, [engine.material(glass_objects[key])])
            engine.transformation(p, trafo((1, 0, 0), (0, 1, 0), (d_, d_, d_), pos_))
            ids["meshes"].append(p)
    if boxes:
        for k, (lo, hi, rot) in enumerate((((-0.72, 0.0, -0.05), (-0.12, 1.2, 0.55), 17.0), ((0.15, 0.0, -0.65), (0.75, 0.6, -0.05), -18.0))):
            if box_lift:
                lo, hi = (lo[0], lo[1] + box_lift, lo[2]), (hi[0], hi[1] + box_lift, hi[2])
            idx, pos, nrm, uv = box_mesh(lo, hi, rot, subdivide=box_subdivide, first_face=box_first_face)
            shape = engine.triangle_mesh(idx, pos, nrm, uv)
            box_mat = engine.material(glass_objects["box"]) if 1 == k and "box" in glass_objects else pick(("tall", "short")[k], white)
            if masked and 0 == k:
                box_mat = engine.material(with_mask(substitute(WHITE), mask_id, sampler=mask_kw["sampler"], scale=[2.0, 2.0]))
            p = engine.prop(shape, [box_mat])
            engine.transformation(p, identity)
            ids["meshes"].append(p)
    for k in range(extra_props):  # plaques a hair in front of the back wall, facing the camera
        p = engine.prop(RECTANGLE, [red if k % 2 else green])
        engine.transformation(p, trafo((-1, 0, 0), (0, 1, 0), (0.18, 0.18, 1), (-0.8 + 1.6 * (k % 6) / 5.0, 0.3 + 0.3 * (k // 6), 0.995)))
        ids["walls"].append(p)
    if mesh is not None:
        idx, pos, nrm, uv = mesh
        shape = engine.triangle_mesh(idx, pos, nrm, uv)
        mm = white if mesh_material is None else mesh_material
        if isinstance(mm, str):  # a material JSON instead of an id
            mm = engine.material(mm)
        p = engine.prop(shape, [mm])
        engine.transformation(p, trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), tuple(mesh_position)))
        ids["meshes"].append(p)

    engine.transformation(cam, trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1, camera_z)))
    return ids


DISTANT = 3  # built-in Distant shape (resource/manager.zig:36-44)
SPHERE = 6   # built-in Sphere: unit diameter, scaled by the transformation
CANOPY = 0   # infinite: the hemisphere of directions around the prop's z axis
DOME = 4     # infinite: the whole sphere of directions


def _basis_for(n):
    """Orthonormal r0, r1 with r0 x r1 = n."""
    n = np.asarray(n, dtype=np.float64)
    n = n / np.linalg.norm(n)
    a = np.array([1.0, 0.0, 0.0]) if abs(n[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    r0 = np.cross(a, n)
    r0 /= np.linalg.norm(r0)
    r1 = np.cross(n, r0)
    return tuple(r0), tuple(r1)


def sky_tables(elevation_deg=35.0, size=64):
    """A stand-in for the products of the reference's sky bake (sky.zig:172-269), for the tests: the sun's radiance over its disc
    (1024 x 3: bright, limb-darkened, reddened) and a canopy image (size x size x 3, zero outside the disc)."""
    v = np.arange(1024, dtype=np.float64) / 1023.0
    y = 2.0 * v - 1.0
    limb = 1.0 - 0.6 * (1.0 - np.sqrt(np.maximum(0.0, 1.0 - y * y)))
    sun = (2.0e4 * limb[:, None] * np.array([1.0, 0.9, 0.75])[None, :] * (0.8 + 0.2 * v[:, None])).astype(np.float32)
    g = (np.arange(size, dtype=np.float64) + 0.5) / size * 2.0 - 1.0
    dx, dy = np.meshgrid(g, g, indexing="xy")
    r = np.sqrt(dx * dx + dy * dy)
    up = np.cos(np.minimum(r, 1.0) * np.pi / 2.0)
    img = np.stack([0.2 + 0.5 * (1 - up) ** 2, 0.35 + 0.45 * (1 - up) ** 2, 0.8 + 0.1 * (1 - up)], axis=-1)
    img *= (1.0 + 1.5 * np.exp(-((dx - 0.0) ** 2 + (dy + np.cos(np.radians(elevation_deg))) ** 2) * 6.0))[..., None]
    img[r > 1.0 + 3.5 / size] = 0.0
    return sun, img.astype(np.float32)


def outdoor(engine, width, height, spp, integrator="PTMIS", sun_radius=0.05, sun_direction=(0.35, 0.8, -0.45), material="lambert",
            extra_quad_light=False, max_depth=8, sky=None, sun=True, showcase=False, mesh=None, sky_map=None, sky_samples=1,
            sky_spectrum=(0.35, 0.5, 0.8), sky_entity=None, shadow_catcher=False, alpha_transparency=False, box_lift=0.0,
            mesh_position=(0.0, 0.01, -1.6)):
    """An open scene for infinite lights: a large ground Rectangle, the two Cornell boxes and a Distant light ("sun",
    distant.zig) whose disc has angular radius atan(sun_radius); optionally one finite quad light next to it, so that the
    light tree mixes the infinite-light distribution with the BVH.  `sky`: "dome" (dome.zig, the whole sphere of
    directions) or "canopy" (canopy.zig, the hemisphere above the ground) with uniform emission; `sun=False` drops the
    Distant light.  showcase: the materials of BASELINE.json's configs[4] -- the red box turns into green glass, a glass
    Sphere and a gold (metal_preset) Sphere stand on the ground; `mesh`: optional (idx, pos, nrm, uv) in polished iron."""
    mk = lambert if material == "lambert" else substitute
    cam = engine.camera(width, height, fov_deg=45.0)
    engine.samples_per_pixel(spp)
    engine.integrators(json.dumps({"surface": {integrator: {"depth": {"surface": max_depth}}}}))
    engine.sensor(json.dumps({"alpha_transparency": True} if alpha_transparency else {}))
    white = engine.material(mk(WHITE))
    red = engine.material(mk(RED))
    sun_mat = engine.material(light_material(spectrum=(12.0, 11.0, 9.0)))
    ids = {"camera": cam}

    ground = engine.prop(RECTANGLE, [white])
    engine.transformation(ground, trafo((1, 0, 0), (0, 0, -1), (12, 12, 1), (0, 0, 0)))
    if shadow_catcher:
        engine.shadow_catcher(ground)
    ids["ground"] = ground
    identity = trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 0))
    if showcase:
        red = engine.material(glass(1.5, (0.55, 0.9, 0.6), 0.4))
        for pos_, d_, mat_ in (((-1.3, 0.35, -0.9), 0.7, glass(1.45)),
                               ((1.4, 0.3, -0.4), 0.6, json.dumps({"rendering": {"Substitute": {"metal_preset": "Gold", "roughness": 0.25}}}))):
            p = engine.prop(SPHERE, [engine.material(mat_)])
            engine.transformation(p, trafo((1, 0, 0), (0, 1, 0), (d_, d_, d_), pos_))
    if mesh is not None:
        iron = engine.material(json.dumps({"rendering": {"Substitute": {"metal_preset": "Iron", "roughness": 0.1}}}))
        p = engine.prop(engine.triangle_mesh(*mesh), [iron])
        engine.transformation(p, trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), mesh_position))
    for lo, hi, rot, mat in (((-0.72, 0.0, -0.05), (-0.12, 1.2, 0.55), 17.0, white), ((0.15, 0.0, -0.65), (0.75, 0.6, -0.05), -18.0, red)):
        if box_lift:  # (see cornell's box_lift)
            lo, hi = (lo[0], lo[1] + box_lift, lo[2]), (hi[0], hi[1] + box_lift, hi[2])
        idx, pos, nrm, uv = box_mesh(lo, hi, rot)
        p = engine.prop(engine.triangle_mesh(idx, pos, nrm, uv), [mat])
        engine.transformation(p, identity)

    if sky_entity is not None:
        tables = sky_entity.get("tables")
        ids["sun"] = engine.sky(sky_entity.get("parameters"), *(tables if tables else (None, None)))
        sun, sky = False, None
    if sun:
        to_light = np.asarray(sun_direction, dtype=np.float64)
        r0, r1 = _basis_for(-to_light)  # the light's z axis points from the light into the scene
        s = engine.prop(DISTANT, [sun_mat])
        engine.transformation(s, trafo(r0, r1, (sun_radius, sun_radius, sun_radius), (0, 0, 0)))
        if shadow_catcher:
            engine.shadow_catcher_light(s)
        engine.light(s)
        ids["sun"] = s
    if sky is not None:
        # sky_map: a numpy image -> image light (emission map x spectrum), importance-sampled
        sky_img = None if sky_map is None else engine.image(sky_map)
        sky_mat = engine.material(light_material(spectrum=sky_spectrum, emission_map=sky_img, num_samples=sky_samples))
        k = engine.prop(DOME if "dome" == sky else CANOPY, [sky_mat])
        # the canopy's z axis is "up": r0 = x, r1 = -z gives r2 = r0 x r1 = +y
        engine.transformation(k, trafo((1, 0, 0), (0, 0, -1), (1, 1, 1), (0, 0, 0)))
        if shadow_catcher:
            engine.shadow_catcher_light(k)
        engine.light(k)
        ids["sky"] = k
    if extra_quad_light:
        lamp = engine.material(light_material(spectrum=(6.0, 6.0, 8.0)))
        q = engine.prop(RECTANGLE, [lamp])
        engine.transformation(q, trafo((1, 0, 0), (0, 0, 1), (0.6, 0.6, 1), (-1.2, 1.6, 0.4)))
        engine.light(q)
    engine.transformation(cam, trafo((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1.1, -4.5)))
    return ids


# ---------------------------------------------------------------- on-disk formats (take / scene JSON, mesh JSON, .sub)

def write_sub_mesh(path, idx, pos, nrm, uv, parts=None, delta=True):
    """A "SUB\\0" binary mesh (reference shape_provider.zig:527-805): u64 JSON size, JSON header, then separate vertex streams
    (positions, normals, uvs) and the index stream, delta-encoded Int32 (every index relative to the one before) or UInt32."""
    import struct
    idx = np.asarray(idx, dtype=np.int64).reshape(-1)
    pos = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    nrm = np.asarray(nrm, dtype=np.float32).reshape(-1, 3)
    uv = np.asarray(uv, dtype=np.float32).reshape(-1, 2)
    vertex_bytes = pos.tobytes() + nrm.tobytes() + uv.tobytes()
    if delta:
        enc = np.diff(np.concatenate([[0], idx])).astype(np.int32).tobytes()
    else:
        enc = idx.astype(np.uint32).tobytes()
    parts = parts or [{"start_index": 0, "num_indices": int(idx.size), "material_index": 0}]
    header = {"geometry": {
        "parts": parts,
        "vertices": {"binary": {"offset": 0, "size": len(vertex_bytes)}, "num_vertices": int(pos.shape[0]),
                     "layout": [{"semantic_name": "Position", "encoding": "Float32x3", "stream": 0},
                                {"semantic_name": "Normal", "encoding": "Float32x3", "stream": 1},
                                {"semantic_name": "Texture_coordinate", "encoding": "Float32x2", "stream": 2}]},
        "indices": {"binary": {"offset": len(vertex_bytes), "size": len(enc)}, "num_indices": int(idx.size),
                    "encoding": "Int32" if delta else "UInt32"}}}
    text = json.dumps(header).encode()
    text += b"\0" * ((8 - len(text) % 8) % 8)
    with open(path, "wb") as f:
        f.write(b"SUB\0" + struct.pack("<Q", len(text)) + text + vertex_bytes + enc)


def write_json_mesh(path, idx, pos, nrm, uv):
    """The JSON mesh format (shape_provider.zig:334-525)."""
    g = {"geometry": {"parts": [{"start_index": 0, "num_indices": int(np.asarray(idx).size), "material_index": 0}],
                      "primitive_topology": "triangle_list",
                      "vertices": {"positions": [float(x) for x in np.asarray(pos, dtype=np.float32).reshape(-1)],
                                   "normals": [float(x) for x in np.asarray(nrm, dtype=np.float32).reshape(-1)],
                                   "texture_coordinates_0": [float(x) for x in np.asarray(uv, dtype=np.float32).reshape(-1)]},
                      "indices": [int(i) for i in np.asarray(idx).reshape(-1)]}}
    with open(path, "w") as f:
        json.dump(g, f)


def write_cornell_take(directory, width, height, spp, integrator="PTMIS", max_depth=16):
    """Writes cornell.take / cornell.scene / the two box meshes (one JSON, one .sub) describing exactly what
    cornell(engine, width, height, spp, material="substitute") builds through the API; returns the take's path."""
    import os

    def m16(r0, r1, scale, position):
        return [float(x) for x in trafo(r0, r1, scale, position).reshape(-1)]

    def mat(name, text):
        d = json.loads(text)
        d["name"] = name
        return d

    materials = [mat("white", substitute(WHITE)), mat("red", substitute(RED)), mat("green", substitute(GREEN)),
                 mat("lamp", light_material())]
    entities = []
    for name, r0, r1, pos in (("white", (1, 0, 0), (0, 0, -1), (0, 0, 0)), ("white", (1, 0, 0), (0, 0, 1), (0, 2, 0)),
                              ("white", (-1, 0, 0), (0, 1, 0), (0, 1, 1)), ("red", (0, 0, -1), (0, 1, 0), (-1, 1, 0)),
                              ("green", (0, 0, 1), (0, 1, 0), (1, 1, 0))):
        entities.append({"type": "Prop", "shape": {"type": "Rectangle"}, "materials": [name],
                         "transformation": m16(r0, r1, (2, 2, 1), pos)})
    for lx in (-0.45, 0.45):
        entities.append({"type": "Light", "shape": {"type": "Rectangle"}, "materials": ["lamp"],
                         "transformation": m16((1, 0, 0), (0, 0, 1), (0.5, 0.5, 1), (lx, 1.98, 0.0))})
    boxes = (((-0.72, 0.0, -0.05), (-0.12, 1.2, 0.55), 17.0), ((0.15, 0.0, -0.65), (0.75, 0.6, -0.05), -18.0))
    write_json_mesh(os.path.join(directory, "tall_box.json"), *box_mesh(*boxes[0]))
    write_sub_mesh(os.path.join(directory, "short_box.sub"), *box_mesh(*boxes[1]))
    for f in ("tall_box.json", "short_box.sub"):
        entities.append({"type": "Prop", "shape": {"file": f}, "materials": ["white"],
                         "transformation": m16((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 0, 0))})
    with open(os.path.join(directory, "cornell.scene"), "w") as f:
        json.dump({"materials": materials, "entities": entities}, f)
    take = {"scene": "cornell.scene",
            "camera": {"Perspective": {"parameters": {"fov": 39.3}, "resolution": [width, height],
                                       "transformation": m16((1, 0, 0), (0, 1, 0), (1, 1, 1), (0, 1, -3.9)),
                                       "shutter": {"motion_blur": False}}},
            "sampler": {"Sobol": {"samples_per_pixel": spp}},
            "integrator": {"surface": {integrator: {"depth": {"surface": max_depth}}}},
            "sensor": {},
            "post": [{"tonemapper": {"Linear": {"exposure": 0.0}}}],
            "export": {"Image": {"format": "PNG"}}}
    path = os.path.join(directory, "cornell.take")
    with open(path, "w") as f:
        json.dump(take, f)
    return path

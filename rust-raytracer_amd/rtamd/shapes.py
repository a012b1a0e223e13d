"""Deterministic synthetic meshes for the configurations the reference ships no data for.

BASELINE.json's C4 asks for a ~100k-triangle OBJ mesh inside the Cornell box; the reference only ships
cube.obj (12 tris) and bun315.obj (4,968 tris, no normals) -- SURVEY.md s0.  `torus` produces an exactly
reproducible mesh of 2*nu*nv triangles with analytic vertex normals (positions/normals/indices as
Mesh::load_obj would deliver them: one normal per vertex, single index)."""
import numpy as np


def torus(nu=160, nv=320, R=1.0, r=0.4):
    """Torus around the y axis: nu segments around the tube, nv around the ring -> 2*nu*nv triangles
    (nu=160, nv=320: 102,400 triangles, 51,200 vertices)."""
    u = (np.arange(nu) / nu) * 2.0 * np.pi
    v = (np.arange(nv) / nv) * 2.0 * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    cx, sx = np.cos(uu), np.sin(uu)
    cv, sv = np.cos(vv), np.sin(vv)
    pos = np.stack([(R + r * cx) * cv, r * sx, (R + r * cx) * sv], axis=-1).reshape(-1, 3)
    nrm = np.stack([cx * cv, sx, cx * sv], axis=-1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j)
    b = (((i + 1) % nu) * nv + j)
    c = (((i + 1) % nu) * nv + (j + 1) % nv)
    d = (i * nv + (j + 1) % nv)
    idx = np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)]).astype(np.uint32)
    # f32-representable coordinates, as tobj would hand them over (mesh.rs:160-172)
    return pos.astype(np.float32).astype(np.float64), nrm.astype(np.float32).astype(np.float64), idx


def sheet(n=8, size=(2.0, 2.0)):
    """A flat n x n grid of quads (2 n^2 triangles) in the plane y = 0, x in [0, size[0]], z in [0, size[1]], normals (0, 1, 0): laid
    exactly on a Cube's face or a rectangle (integer / dyadic coordinates) its triangles share their t with that surface -- bit for bit for a
    fair share of the rays, since both are the same plane -- which is what the exact-tie tests of kernels 5 / 6 need from a mesh instance."""
    g = np.arange(n + 1) / n
    xx, zz = np.meshgrid(g * size[0], g * size[1], indexing="ij")
    pos = np.stack([xx, np.zeros_like(xx), zz], axis=-1).reshape(-1, 3)
    nrm = np.tile(np.array([0.0, 1.0, 0.0]), (pos.shape[0], 1))
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a, b, c, d = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
    idx = np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)]).astype(np.uint32)
    return pos.astype(np.float32).astype(np.float64), nrm, idx


def box_mesh(n=4, size=(1.0, 1.0, 1.0)):
    """A closed axis-aligned box [0, size] as a triangle mesh: six faces of n x n quads (12 n^2 triangles), one normal per face (vertices
    are duplicated along the edges, as an OBJ with per-face normals loads under tobj's single_index)."""
    P, N, I = [], [], []
    g = np.arange(n + 1) / n
    for axis in range(3):
        for side in (0, 1):
            u, v = [a for a in range(3) if a != axis]
            uu, vv = np.meshgrid(g * size[u], g * size[v], indexing="ij")
            pos = np.zeros(uu.shape + (3,))
            pos[..., u], pos[..., v], pos[..., axis] = uu, vv, side * size[axis]
            nrm = np.zeros(3)
            nrm[axis] = 1.0 if side else -1.0
            base = sum(p.shape[0] for p in P)
            P.append(pos.reshape(-1, 3))
            N.append(np.tile(nrm, ((n + 1) ** 2, 1)))
            i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
            a, b, c, d = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
            I.append(base + np.concatenate([np.stack([a, b, c], axis=-1).reshape(-1, 3), np.stack([a, c, d], axis=-1).reshape(-1, 3)]))
    return (np.concatenate(P).astype(np.float32).astype(np.float64), np.concatenate(N), np.concatenate(I).astype(np.uint32))


def cornell_with_mesh(world_or_scene, positions, normals, indices, scale=120.0, translate=(278.0, 200.0, 278.0), rotate=(30.0, 20.0, 0.0),
                      seed=1, mesh_fn=None):
    """scene.rs:16-112 with the cube.obj mesh replaced by a given mesh (C4).  Works on both builders
    (rtamd.World and the oracle's Scene) since they share the reference's constructor names.
    mesh_fn(builder, material) -> mesh object replaces the Mesh(positions, normals, indices) call (e.g. Mesh::load_obj on a file).
    Returns the list of top-level hitables (to be passed to World::new)."""
    w = world_or_scene
    red = w.Lambertian(w.ConstantTexture((0.75, 0.25, 0.25)))
    white = w.Lambertian(w.ConstantTexture((0.75, 0.75, 0.75)))
    blue = w.Lambertian(w.ConstantTexture((0.25, 0.25, 0.75)))
    light = w.DiffuseLight(w.ConstantTexture((1.0, 1.0, 1.0)))
    if mesh_fn is not None:
        mesh = mesh_fn(w, white)
    else:
        try:
            mesh = w.Mesh(positions, normals, indices, white, bvh_seed=seed)
        except TypeError:
            mesh = w.Mesh(positions, normals, indices, white, seed)
    return [
        w.YZRectangle((0.0, 0.0), (555.0, 555.0), 555.0, red),
        w.YZRectangle((0.0, 0.0), (555.0, 555.0), 0.0, blue),
        w.XZRectangle((0.0, 0.0), (555.0, 555.0), 0.0, white),
        w.XZRectangle((0.0, 0.0), (555.0, 555.0), 555.0, white),
        w.XYRectangle((0.0, 0.0), (555.0, 555.0), 555.0, white),
        w.Sphere((140.0, 100.0, 240.0), 100.0, w.Dielectric(1.5, w.ConstantTexture((0.999, 0.999, 0.999)))),
        w.Sphere((400.0, 100.0, 360.0), 100.0, w.Metal(w.ConstantTexture((0.999, 0.999, 0.999)), 0.0)),
        w.XZRectangle((213.0, 227.0), (343.0, 332.0), 554.0, light),
        w.Transform(rotate, (scale, scale, scale), translate, mesh),
        w.Cube((300.0, 0.0, 100.0), (380.0, 100.0, 180.0), white),
    ]


CORNELL_CAMERA = dict(look_from=(278.0, 278.0, -800.0), look_at=(278.0, 278.0, 278.0), vup=(0.0, 1.0, 0.0), vfov=50.0, aperture=0.0,
                      focus_dist=10.0)


def final_scene(B, n_boxes=20, n_cluster=1000, seed=2):
    """BASELINE config C5 AS NAMED: book 2's final scene with its moving sphere (centre (400,400,200) -> (430,400,200) while the shutter
    [0, 1) is open: FINAL_SCENE_SHUTTER) and its Perlin marble sphere (noise_texture(0.1)).  The reference has code for neither (ray.rs:3-6
    has no time; no noise texture): both are book-2 extensions of this build, parity exists against the own restatement only."""
    return final_scene_reduced(B, n_boxes, n_cluster, seed, full=True)


FINAL_SCENE_SHUTTER = (0.0, 1.0)


def final_scene_reduced(B, n_boxes=20, n_cluster=1000, seed=2, full=False):
    """BASELINE config C5 ("motion-blur + Perlin-noise volumetric final scene", book 2) reduced to what the reference has code for:
    no motion blur (ray.rs:3-6 has no time: the moving sphere stands still) and no Perlin noise (no noise texture: that sphere gets a
    checker).  Everything else is the book's scene: a ground of n_boxes^2 boxes of random height, a rectangle light, glass and metal
    spheres, a blue subsurface ball (a ConstantMedium inside a glass sphere), a thin global fog (a ConstantMedium in a sphere of
    radius 5000), an image-textured sphere, and n_cluster small spheres in their own BVH under a rotate + translate Transform.
    Works on both builders (rtamd.World and the oracle's Scene).  Returns the list of top-level hitables."""
    rng = np.random.default_rng(seed)
    ground = B.Lambertian(B.ConstantTexture((0.48, 0.83, 0.53)))
    items = []
    w = 100.0 * 20.0 / n_boxes
    boxes = []
    for i in range(n_boxes):
        for j in range(n_boxes):
            x0, z0 = -1000.0 + i * w, -1000.0 + j * w
            boxes.append(B.Cube((x0, 0.0, z0), (x0 + w, float(rng.uniform(1.0, 101.0)), z0 + w), ground))
    items.append(B.BVHNode_new(boxes, 11))
    items.append(B.XZRectangle((123.0, 147.0), (423.0, 412.0), 554.0, B.DiffuseLight(B.ConstantTexture((7.0, 7.0, 7.0)))))
    if full:
        items.append(B.MovingSphere((400.0, 400.0, 200.0), (430.0, 400.0, 200.0), 0.0, 1.0, 50.0, B.Lambertian(B.ConstantTexture((0.7, 0.3, 0.1)))))
    else:
        items.append(B.Sphere((400.0, 400.0, 200.0), 50.0, B.Lambertian(B.ConstantTexture((0.7, 0.3, 0.1)))))          # the "moving" sphere
    items.append(B.Sphere((260.0, 150.0, 45.0), 50.0, B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0)))))
    items.append(B.Sphere((0.0, 150.0, 145.0), 50.0, B.Metal(B.ConstantTexture((0.8, 0.8, 0.9)), 1.0)))
    glass = B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0)))
    items.append(B.Sphere((360.0, 150.0, 145.0), 70.0, glass))
    items.append(B.ConstantMedium(0.2, B.Sphere((360.0, 150.0, 145.0), 70.0, glass), B.Isotropic(B.ConstantTexture((0.2, 0.4, 0.9)))))
    items.append(B.ConstantMedium(0.0001, B.Sphere((0.0, 0.0, 0.0), 5000.0, glass), B.Isotropic(B.ConstantTexture((1.0, 1.0, 1.0)))))
    yy, xx = np.mgrid[0:64, 0:128]
    earth = np.stack([(40 + 150 * (np.sin(xx / 9.0) * np.cos(yy / 7.0) > 0.2)), 90 + (xx * 3 + yy * 5) % 120, 160 + (yy * 11) % 90], axis=-1).astype(np.uint8)
    items.append(B.Sphere((400.0, 200.0, 400.0), 100.0, B.Lambertian(B.ImageTexture(earth))))
    if full:
        items.append(B.Sphere((220.0, 280.0, 300.0), 80.0, B.Lambertian(B.NoiseTexture(0.1, 5))))
    else:
        items.append(B.Sphere((220.0, 280.0, 300.0), 80.0, B.Lambertian(B.CheckerTexture(B.ConstantTexture((0.2, 0.2, 0.2)), B.ConstantTexture((0.9, 0.9, 0.9))))))  # (Perlin in the book)
    white = B.Lambertian(B.ConstantTexture((0.73, 0.73, 0.73)))
    cluster = [B.Sphere(tuple(float(c) for c in rng.uniform(0.0, 165.0, 3)), 10.0, white) for _ in range(n_cluster)]
    items.append(B.Transform((0.0, 15.0, 0.0), (1.0, 1.0, 1.0), (-100.0, 270.0, 395.0), B.BVHNode_new(cluster, 12)))
    return items


FINAL_SCENE_CAMERA = ((478.0, 278.0, -600.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0)

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

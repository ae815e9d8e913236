"""Uniform area-weighted surface sampling (pytorch3d.ops.sample_points_from_meshes interface)."""
import torch


def sample_points_from_meshes(meshes, num_samples: int = 10000, return_normals: bool = False, return_textures: bool = False):
    if meshes.isempty():
        raise ValueError("Meshes are empty.")
    verts = meshes.verts_packed()
    if not torch.isfinite(verts).all():
        raise ValueError("Meshes contain nan or inf.")
    faces = meshes.faces_packed()
    mesh_to_face = meshes.mesh_to_faces_packed_first_idx()
    num_faces = meshes.num_faces_per_mesh()
    N = len(meshes)
    samples = torch.zeros((N, num_samples, 3), device=meshes.device)
    normals = torch.zeros((N, num_samples, 3), device=meshes.device)
    areas = meshes.faces_areas_packed()
    fnormals = meshes.faces_normals_packed() if return_normals else None
    for n in range(N):
        f0, nf = int(mesh_to_face[n]), int(num_faces[n])
        if nf == 0:
            continue
        a = areas[f0:f0 + nf]
        with torch.no_grad():
            fi = torch.multinomial(a.clamp(min=0) + 1e-30, num_samples, replacement=True) + f0
        v0, v1, v2 = verts[faces[fi, 0]], verts[faces[fi, 1]], verts[faces[fi, 2]]
        u = torch.rand(2, num_samples, dtype=verts.dtype, device=verts.device)
        su = u[0].sqrt()
        w0, w1, w2 = 1.0 - su, su * (1.0 - u[1]), su * u[1]
        samples[n] = w0[:, None] * v0 + w1[:, None] * v1 + w2[:, None] * v2
        if return_normals:
            normals[n] = fnormals[fi]
    if return_normals:
        return samples, normals
    return samples

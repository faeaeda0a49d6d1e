/*
 * bibim_assets.h -- asset ingestion for the forward path: the step BEFORE the path.
 * Plain C ABI, host only (no HIP): the bytes these functions return are what bbr_upload_mesh / bbr_upload_material
 * take.  They replace the reference's use of assimp and stb_image for the three asset kinds the path consumes:
 *
 *   ShaderBall.fbx  -> bb::Vertex[]        src/scene.cpp:57-86   (assimp, aiProcess_Triangulate | CalcTangentSpace)
 *   gizmo.obj/.mtl  -> bb::GizmoVertex[]   src/main.cpp:219-283  (assimp, aiProcess_Triangulate)
 *   pbr/<name>/{albedo,...}.png -> RGBA8     src/render.cpp:1243-1316, src/resource.cpp:176-216 (stbi_load, STBI_rgb_alpha)
 *
 * assimp is a binary-only dependency of the reference (Windows import libraries, no source, DLLs absent), so the FBX
 * and OBJ rules are the ones SURVEY.md 8(c) states (parity unpinned against assimp itself; pinned against the
 * committed fixtures made by tools/fbx_geometry.py and tools/obj_loader.py).  The PNG decoder is pinned against the
 * reference's own stb_image 2.25 (compiled in the authoring container, oracle/_ref) on every PNG the reference ships.
 *
 * All returned buffers are malloc'ed; release them with bba_free.  Functions return 0 on success, a negative
 * bba_status otherwise and leave a message in bba_last_error() (thread-local).
 */
#ifndef BIBIM_ASSETS_H
#define BIBIM_ASSETS_H

#include <stdint.h>

#include "bibim_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bba_status {
  BBA_OK = 0,
  BBA_ERR_IO = -1,          /* cannot open / read */
  BBA_ERR_FORMAT = -2,      /* not the expected file format, or corrupt */
  BBA_ERR_UNSUPPORTED = -3, /* valid file using a feature outside the path's needs (layer mappings other than ByPolygonVertex, ...) */
  BBA_ERR_ARGUMENT = -4
} bba_status;

const char *bba_last_error(void);
void bba_free(void *p);

/* Binary FBX 7.x (32-bit records, or the 64-bit records of 7500 and later): the first Geometry object expanded to the non-indexed triangle list
 * ShaderBallScene builds -- vertex k = (Vertices[PolygonVertexIndex[k] (decoded)], UV0[UVIndex0[k]], Normals[k],
 * Tangents[k]), doubles cast to float, file tangents kept, no UV flip, no unit scaling.  Polygons with more than three
 * corners are fanned from their first corner (aiProcess_Triangulate on convex polygons).  out_vertices: 44-byte
 * bb::Vertex records. */
int bba_load_fbx_vertices(const char *path, void **out_vertices, uint32_t *out_n_vertices);

/* Wavefront OBJ + MTL as the gizmo loader uses it: one vertex per face corner (position, Kd of the face's material,
 * normal), polygons fan-triangulated from their first corner, file order.  out_vertices: 36-byte records
 * (pos[3], color[3], normal[3]); out_indices: three per triangle. */
int bba_load_obj_gizmo(const char *path, void **out_vertices, uint32_t *out_n_vertices, uint32_t **out_indices,
                       uint32_t *out_n_indices);

/* PNG -> RGBA8 with the conversions of stbi_load(..., STBI_rgb_alpha): grey -> (y,y,y,255), 16-bit -> high byte,
 * 1/2/4-bit grey scaled to 0..255, palette expanded, tRNS honoured, Adam7 interlace supported. */
int bba_load_png(const char *path, uint8_t **out_rgba, int32_t *out_width, int32_t *out_height);
int bba_decode_png(const uint8_t *bytes, uint64_t n_bytes, uint8_t **out_rgba, int32_t *out_width, int32_t *out_height);

/* One material directory: albedo.png metallic.png roughness.png ao.png normal.png height.png, each optional (a missing
 * file = NULL map = the `default` material's map, src/render.cpp:1328-1336); uploads it to `ctx`. */
int bba_load_material_dir(bbr_context *ctx, const char *dir, int32_t *out_material);

/* createPBRMaterialSet (src/render.cpp:1243-1316): every sub-directory of `pbr_root` in name order; the one called
 * "default" is swapped with the last and dropped (its maps are what missing maps fall back to -- built into the
 * library).  Writes up to `capacity` material handles and their directory names (64 bytes each, NUL-terminated). */
int bba_load_material_set(bbr_context *ctx, const char *pbr_root, int32_t *out_materials, char (*out_names)[64],
                          uint32_t capacity, uint32_t *out_n);

#ifdef __cplusplus
}
#endif
#endif

//go:build hip

// NOT COMPILED HERE (no Go toolchain in the image this library is built in).  The shape half of
// flattenScene: every Shape becomes {rt_shape_kind, index into the array of that kind}.
package phys

/*
#include "rt_hip.h"
*/
import "C"

import "fmt"

func (f *flatScene) triangle(kind C.int32_t, p [3][3]float64, uv [3][2]float64, n [3]float64) (C.int32_t, int32) {
	var t C.rt_triangle
	t.kind = kind
	for i := 0; i < 3; i++ {
		put3(&t.p[i], p[i][0], p[i][1], p[i][2])
		t.uv[i][0], t.uv[i][1] = C.double(uv[i][0]), C.double(uv[i][1])
	}
	put3(&t.normal, n[0], n[1], n[2])
	f.tris = append(f.tris, t)
	return C.RT_SHAPE_TRIANGLE, int32(len(f.tris) - 1)
}

func (f *flatScene) shape(s Shape) (C.int32_t, int32, error) {
	switch sh := s.(type) {
	case Sphere: // shape_sphere.go:245-256
		c := C.rt_sphere{radius: C.double(sh.Radius), uvmap: C.int32_t(sh.UVMap)} // UVMapKind values are rt_uvmap_kind's
		put3(&c.center, sh.Center.X, sh.Center.Y, sh.Center.Z)
		f.sph = append(f.sph, c)
		return C.RT_SHAPE_SPHERE, int32(len(f.sph) - 1), nil
	case Quad: // shape_quad.go:15-20; the library derives the corners as Collide does (:39-76)
		c := C.rt_quad{width: C.double(sh.Width), height: C.double(sh.Height)}
		put3(&c.center, sh.Center.X, sh.Center.Y, sh.Center.Z)
		put3(&c.normal, sh.Normal.X, sh.Normal.Y, sh.Normal.Z)
		f.quads = append(f.quads, c)
		return C.RT_SHAPE_QUAD, int32(len(f.quads) - 1), nil
	case Cylinder: // shape_cylinder.go:13-18
		c := C.rt_cylinder{radius: C.double(sh.Radius), height: C.double(sh.Height)}
		put3(&c.origin, sh.Origin.X, sh.Origin.Y, sh.Origin.Z)
		put3(&c.direction, sh.Direction.X, sh.Direction.Y, sh.Direction.Z)
		f.cyls = append(f.cyls, c)
		return C.RT_SHAPE_CYLINDER, int32(len(f.cyls) - 1), nil
	case Triangle:
		k, i := f.triangle(C.RT_TRI_TRIANGLE, [3][3]float64{{sh.P0.X, sh.P0.Y, sh.P0.Z}, {sh.P1.X, sh.P1.Y, sh.P1.Z}, {sh.P2.X, sh.P2.Y, sh.P2.Z}},
			[3][2]float64{}, [3]float64{})
		return k, i, nil
	case TriangleUV:
		k, i := f.triangle(C.RT_TRI_TRIANGLE_UV, [3][3]float64{{sh.P0.X, sh.P0.Y, sh.P0.Z}, {sh.P1.X, sh.P1.Y, sh.P1.Z}, {sh.P2.X, sh.P2.Y, sh.P2.Z}},
			[3][2]float64{{sh.UV0.X, sh.UV0.Y}, {sh.UV1.X, sh.UV1.Y}, {sh.UV2.X, sh.UV2.Y}}, [3]float64{sh.Normal.X, sh.Normal.Y, sh.Normal.Z})
		return k, i, nil
	case Face: // a lone Face as a node's shape: eps 1e-8, barycentric uv (shape_mesh.go:95-142)
		var p [3][3]float64
		var uv [3][2]float64
		for i, v := range sh.Vertex {
			p[i], uv[i] = [3]float64{v.Position.X, v.Position.Y, v.Position.Z}, [2]float64{v.UV.X, v.UV.Y}
		}
		k, i := f.triangle(C.RT_TRI_FACE, p, uv, [3]float64{})
		return k, i, nil
	case *Mesh: // the Go-side tree (sh.BVH) is not needed: the library rebuilds it by NewBVH's rules
		f.meshes = append(f.meshes, C.rt_mesh{first_face: C.uint64_t(len(f.faces)), n_faces: C.uint64_t(len(sh.Face))})
		f.faces = append(f.faces, sh.Face...)
		return C.RT_SHAPE_MESH, int32(len(f.meshes) - 1), nil
	case TransformedShape: // shape_transformed_shape.go:12-15; the library inverts the matrix itself
		ck, ci, err := f.shape(sh.Shape)
		if err != nil {
			return 0, 0, err
		}
		c := C.rt_transformed{child_kind: ck, child_index: C.int32_t(ci)}
		t := sh.Transform
		put3(&c.translation, t.Translation.X, t.Translation.Y, t.Translation.Z)
		put3(&c.scale, t.Scale.X, t.Scale.Y, t.Scale.Z)
		for i := 0; i < 9; i++ {
			c.rotation[i] = C.double(t.Rotation.M[i/3][i%3])
		}
		f.xf = append(f.xf, c)
		return C.RT_SHAPE_TRANSFORMED, int32(len(f.xf) - 1), nil
	}
	return 0, 0, fmt.Errorf("shape type %T has no GPU mapping", s)
}

// flattenScene fills rt_scene_desc from scene.Node and scene.Light.  The arrays live in C memory
// until free(); rt_scene_create deep-copies them, so the caller may free right after it.
func flattenScene(scene *Scene) (*flatScene, error) {
	f := &flatScene{texOf: map[Texture]int32{}}
	matOf := map[Material]int32{}
	for i, n := range scene.Node {
		kind, idx, err := f.shape(n.Shape)
		if err != nil {
			f.free()
			return nil, fmt.Errorf("node %d (%s): %w", i, n.Name, err)
		}
		m, seen := matOf[n.Material]
		if !seen {
			if m, err = f.material(n.Material); err != nil {
				f.free()
				return nil, fmt.Errorf("node %d (%s): %w", i, n.Name, err)
			}
			matOf[n.Material] = m
		}
		f.nodes = append(f.nodes, C.rt_node{shape_kind: kind, shape_index: C.int32_t(idx), material: C.int32_t(m)})
	}
	for i, l := range scene.Light {
		pl, ok := l.(PointLight) // light.go:16-26, the only Light the reference defines
		if !ok {
			f.free()
			return nil, fmt.Errorf("light %d: type %T has no GPU mapping", i, l)
		}
		var c C.rt_light
		put3(&c.position, pl.Position.X, pl.Position.Y, pl.Position.Z)
		put3(&c.intensity, pl.RadiantIntensity.X, pl.RadiantIntensity.Y, pl.RadiantIntensity.Z)
		f.lights = append(f.lights, c)
	}
	d := &f.desc
	d.nodes, d.n_nodes = (*C.rt_node)(toC(f, f.nodes)), C.uint64_t(len(f.nodes))
	d.spheres, d.n_spheres = (*C.rt_sphere)(toC(f, f.sph)), C.uint64_t(len(f.sph))
	d.quads, d.n_quads = (*C.rt_quad)(toC(f, f.quads)), C.uint64_t(len(f.quads))
	d.cylinders, d.n_cylinders = (*C.rt_cylinder)(toC(f, f.cyls)), C.uint64_t(len(f.cyls))
	d.triangles, d.n_triangles = (*C.rt_triangle)(toC(f, f.tris)), C.uint64_t(len(f.tris))
	d.meshes, d.n_meshes = (*C.rt_mesh)(toC(f, f.meshes)), C.uint64_t(len(f.meshes))
	d.faces, d.n_faces = (*C.rt_face)(toC(f, f.faces)), C.uint64_t(len(f.faces)) // one memcpy of len*120 bytes
	d.transformed, d.n_transformed = (*C.rt_transformed)(toC(f, f.xf)), C.uint64_t(len(f.xf))
	d.materials, d.n_materials = (*C.rt_material)(toC(f, f.mats)), C.uint64_t(len(f.mats))
	d.textures, d.n_textures = (*C.rt_texture)(toC(f, f.texs)), C.uint64_t(len(f.texs))
	d.images, d.n_images = (*C.rt_image)(toC(f, f.imgs)), C.uint64_t(len(f.imgs))
	d.lights, d.n_lights = (*C.rt_light)(toC(f, f.lights)), C.uint64_t(len(f.lights))
	return f, nil
}

//go:build hip

// NOT COMPILED HERE (no Go toolchain in the image this library is built in).  flattenScene turns
// a *Scene into the arrays of rt_scene_desc: one pass over scene.Node in order (tracePath visits
// nodes in order, render.go:155-169), a type switch per shape / material / texture.
package phys

/*
#include <stdlib.h>
#include <string.h>
#include "rt_hip.h"
*/
import "C"

import (
	"fmt"
	"image"
	"image/draw"
	"unsafe"

	"github.com/scottlawsonbc/slam/code/photon/raytrace/r3"
)

func put3(dst *[3]C.double, x, y, z float64) { dst[0], dst[1], dst[2] = C.double(x), C.double(y), C.double(z) }

func lookAt(d *C.rt_camera_desc, from, at r3.Point, up r3.Vec, fovH, fovW Distance) {
	put3(&d.look_from, from.X, from.Y, from.Z)
	put3(&d.look_at, at.X, at.Y, at.Z)
	put3(&d.vup, up.X, up.Y, up.Z)
	d.fov_height, d.fov_width = C.double(fovH), C.double(fovW)
}

type flatScene struct {
	desc   C.rt_scene_desc
	nodes  []C.rt_node
	sph    []C.rt_sphere
	quads  []C.rt_quad
	cyls   []C.rt_cylinder
	tris   []C.rt_triangle
	meshes []C.rt_mesh
	faces  []Face // rt_face has the memory layout of Face (shape_mesh.go:12-21): no conversion
	xf     []C.rt_transformed
	mats   []C.rt_material
	texs   []C.rt_texture
	imgs   []C.rt_image
	lights []C.rt_light
	texOf  map[Texture]int32
	cmem   []unsafe.Pointer // C copies handed to the library; released by free()
}

func (f *flatScene) free() {
	for _, p := range f.cmem {
		C.free(p)
	}
}

// toC copies a Go slice into C memory (cgo must not be handed Go memory that other Go pointers
// reach); the library deep-copies everything inside rt_scene_create, so free() may follow at once.
func toC[T any](f *flatScene, s []T) unsafe.Pointer {
	if len(s) == 0 {
		return nil
	}
	n := C.size_t(len(s)) * C.size_t(unsafe.Sizeof(s[0]))
	p := C.malloc(n)
	C.memcpy(p, unsafe.Pointer(&s[0]), n)
	f.cmem = append(f.cmem, p)
	return p
}

func (f *flatScene) texture(t Texture) (int32, error) {
	if i, ok := f.texOf[t]; ok {
		return i, nil
	}
	var c C.rt_texture
	switch tx := t.(type) {
	case TextureUniform:
		c.kind = C.RT_TEX_UNIFORM
		put3(&c.color, tx.Color.X, tx.Color.Y, tx.Color.Z)
	case TextureCheckerboard:
		odd, err := f.texture(tx.Odd)
		if err != nil {
			return 0, err
		}
		even, err := f.texture(tx.Even)
		if err != nil {
			return 0, err
		}
		c.kind, c.odd, c.even, c.frequency = C.RT_TEX_CHECKERBOARD, C.int32_t(odd), C.int32_t(even), C.double(tx.Frequency)
	case TextureImage: // texture_image.go:60-151: "bilinear" / "clamp", anything else = nearest / repeat
		b := tx.Image.Bounds()
		px := image.NewNRGBA(image.Rect(0, 0, b.Dx(), b.Dy()))
		draw.Draw(px, px.Bounds(), tx.Image, b.Min, draw.Src) // 16-bit sources: fill an RGBA64 and use RT_IMG_RGBA16 instead
		f.imgs = append(f.imgs, C.rt_image{width: C.int32_t(b.Dx()), height: C.int32_t(b.Dy()), format: C.RT_IMG_NRGBA8, pixels: toC(f, px.Pix)})
		c.kind, c.image = C.RT_TEX_IMAGE, C.int32_t(len(f.imgs)-1)
		if tx.Interp == "bilinear" {
			c.interp = C.RT_INTERP_BILINEAR
		}
		if tx.WrapMode == "clamp" {
			c.wrap = C.RT_WRAP_CLAMP
		}
	default:
		return 0, fmt.Errorf("texture type %T has no GPU mapping", t)
	}
	f.texs = append(f.texs, c)
	f.texOf[t] = int32(len(f.texs) - 1)
	return f.texOf[t], nil
}

func (f *flatScene) material(m Material) (int32, error) {
	var c C.rt_material
	var tex Texture
	switch mt := m.(type) {
	case Lambertian:
		c.kind, tex = C.RT_MAT_LAMBERTIAN, mt.Texture
	case Emitter:
		c.kind, tex = C.RT_MAT_EMITTER, mt.Texture
	case Metal:
		c.kind, c.fuzz = C.RT_MAT_METAL, C.double(mt.Fuzz)
		put3(&c.albedo, mt.Albedo.X, mt.Albedo.Y, mt.Albedo.Z)
	case Dielectric:
		c.kind = C.RT_MAT_DIELECTRIC
		c.ior_interior, c.ior_exterior, c.roughness = C.double(mt.RefractiveIndexInterior), C.double(mt.RefractiveIndexExterior), C.double(mt.Roughness)
	case DebugUV:
		c.kind = C.RT_MAT_DEBUG_UV
	case DebugNormal:
		c.kind = C.RT_MAT_DEBUG_NORMAL
	default:
		return 0, fmt.Errorf("material type %T has no GPU mapping", m)
	}
	if tex != nil {
		i, err := f.texture(tex)
		if err != nil {
			return 0, err
		}
		c.texture = C.int32_t(i)
	}
	f.mats = append(f.mats, c)
	return int32(len(f.mats) - 1), nil
}

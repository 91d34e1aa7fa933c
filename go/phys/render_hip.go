//go:build hip

// NOT COMPILED HERE: the image this library is built in has no Go toolchain.  This file is the
// reference-side half of the boundary, written against include/rt_hip.h; the same calls are
// exercised through ctypes by raytrace_amd/phys.py.  A maintainer drops it into phys/ and tags
// the existing renderScene (render.go:285-349) with `//go:build !hip`.
package phys

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raytrace_amd -lrt_hip -Wl,-rpath,${SRCDIR}/../../raytrace_amd
#include "rt_hip.h"
*/
import "C"

import (
	"context"
	"fmt"
	"image"
	"os"
	"runtime"
	"strconv"
	"strings"
	"time"
	"unsafe"
)

// RT_DEVICES="0,1,2,3" renders every frame on those GPUs (rows or samples split inside the
// library); unset = device 0.  There is no CPU fallback: without a GPU the package panics.
func init() {
	var ids []C.int
	for _, f := range strings.Split(os.Getenv("RT_DEVICES"), ",") {
		if v, err := strconv.Atoi(strings.TrimSpace(f)); err == nil {
			ids = append(ids, C.int(v))
		}
	}
	var p *C.int
	if len(ids) > 0 {
		p = &ids[0]
	}
	// rt_last_error() is thread local: the call and the read of its message must run on one OS thread
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := C.rt_init(p, C.int(len(ids))); rc != C.RT_OK {
		panic("rt_init: " + C.GoString(C.rt_last_error()))
	}
}

// cameraDesc maps the four concrete cameras onto rt_camera_desc; an AnimatedCamera is resolved
// first, as its Cast does on every ray (camera_animated.go:70-74).
func cameraDesc(c Camera) (C.rt_camera_desc, error) {
	var d C.rt_camera_desc
	if a, ok := c.(AnimatedCamera); ok {
		c = a.Build(a.U)
	}
	switch cam := c.(type) {
	case PinholeCamera:
		d.kind = C.RT_CAM_PINHOLE
		put3(&d.lower_left_corner, cam.LowerLeftCorner.X, cam.LowerLeftCorner.Y, cam.LowerLeftCorner.Z)
		put3(&d.origin, cam.Origin.X, cam.Origin.Y, cam.Origin.Z)
		put3(&d.horizontal, cam.Horizontal.X, cam.Horizontal.Y, cam.Horizontal.Z)
		put3(&d.vertical, cam.Vertical.X, cam.Vertical.Y, cam.Vertical.Z)
	case FocusableCamera:
		d.kind = C.RT_CAM_FOCUSABLE
		lookAt(&d, cam.LookFrom, cam.LookAt, cam.VUp, cam.FOVHeight, cam.FOVWidth)
		d.aperture, d.working_distance = C.double(cam.Aperture), C.double(cam.WorkingDistance)
	case OrthographicCamera:
		d.kind = C.RT_CAM_ORTHOGRAPHIC
		lookAt(&d, cam.LookFrom, cam.LookAt, cam.VUp, cam.FOVHeight, cam.FOVWidth)
	case CalibratedCamera:
		d.kind = C.RT_CAM_CALIBRATED
		lookAt(&d, cam.Extrinsics.LookFrom, cam.Extrinsics.LookAt, cam.Extrinsics.VUp, 0, 0)
		in := cam.Intrinsics
		d.width, d.height = C.int32_t(in.Width), C.int32_t(in.Height)
		d.fx, d.fy, d.cx, d.cy = C.double(in.Fx), C.double(in.Fy), C.double(in.Cx), C.double(in.Cy)
		d.k1, d.k2, d.p1, d.p2 = C.double(in.K1), C.double(in.K2), C.double(in.P1), C.double(in.P2)
		d.k3, d.k4, d.k5, d.k6 = C.double(in.K3), C.double(in.K4), C.double(in.K5), C.double(in.K6)
	default:
		return d, fmt.Errorf("camera type %T has no GPU mapping", c)
	}
	return d, nil
}

// renderScene replaces render.go:285-349: same arguments, same artifact, same cancellation.
func renderScene(ctx context.Context, scene *Scene, camera Camera) (RenderArtifact, error) {
	t0 := time.Now()
	flat, err := flattenScene(scene) // flatten_hip.go
	if err != nil {
		return RenderArtifact{}, err
	}
	defer flat.free()
	cam, err := cameraDesc(camera)
	if err != nil {
		return RenderArtifact{}, err
	}
	var h *C.rt_scene
	runtime.LockOSThread() // rt_last_error() is thread local
	rc := C.rt_scene_create(&flat.desc, &h)
	msg := ""
	if rc != C.RT_OK {
		msg = C.GoString(C.rt_last_error())
	}
	runtime.UnlockOSThread()
	if rc != C.RT_OK {
		return RenderArtifact{}, fmt.Errorf("rt_scene_create: %s", msg)
	}
	defer C.rt_scene_destroy(h) // a frame loop keeps h and changes only the camera
	o := scene.RenderOptions
	opts := C.rt_render_opts{seed: C.int64_t(o.Seed), rays_per_pixel: C.int32_t(o.RaysPerPixel),
		max_ray_depth: C.int32_t(o.MaxRayDepth), dx: C.int32_t(o.Dx), dy: C.int32_t(o.Dy),
		mode: C.RT_MODE_FAST, band_rank: 0, band_world: 1, device_split: C.RT_SPLIT_AUTO}
	img := image.NewRGBA(image.Rect(0, 0, o.Dx, o.Dy))
	var st C.rt_stats
	type result struct {
		rc  C.int
		msg string
	}
	done := make(chan result, 1)
	go func() {
		runtime.LockOSThread() // rt_last_error() is thread local
		defer runtime.UnlockOSThread()
		rc := C.rt_render(h, &cam, &opts, (*C.uint8_t)(unsafe.Pointer(&img.Pix[0])), C.size_t(img.Stride), &st)
		done <- result{rc, C.GoString(C.rt_last_error())}
	}()
	var r result
	select {
	case <-ctx.Done(): // render.go:340-343
		C.rt_cancel(h)
		<-done
		return RenderArtifact{}, ctx.Err()
	case r = <-done:
	}
	if r.rc != C.RT_OK {
		return RenderArtifact{}, fmt.Errorf("rt_render: %s", r.msg)
	}
	return RenderArtifact{Image: img, Stats: RenderStats{Dx: o.Dx, Dy: o.Dy, TotalRays: uint64(st.total_rays),
		RaysExceededDepth: uint64(st.rays_exceeded_depth), RaysLeftScene: uint64(st.rays_left_scene),
		RenderTime: time.Since(t0)}}, nil
}

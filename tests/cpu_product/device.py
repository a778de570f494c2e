"""TEST INFRASTRUCTURE ONLY.  The "device" of the CPU build of the product's host code (tests/cpu_product/stub_device.cpp): every kernel launcher of libmifx hands its arguments
to launch() below, which runs the REFERENCE's shader for that pass (oracle/_ref) on the planes it was given -- host memory, since the stand-in HIP runtime allocates with calloc.
What runs is therefore the product's own sequencing (csrc/api_*.cpp: which planes, which ping-pong slot, which reset flag and alpha, what is cleared when) around the reference's
arithmetic; tests compare its outputs with oracle/cpu_chain.py, whose sequencing is pinned to the executed reference host classes (tests/test_host_sequence_vs_ref.py): any
difference is a difference of sequencing.  Nothing under diligentfx_amd/ imports this."""
import ctypes
import threading

import numpy as np

import cpu_chain


class Img(ctypes.Structure):  # mifx::Img (mifx_device.h)
    _fields_ = [("p", ctypes.c_void_p), ("w", ctypes.c_int), ("h", ctypes.c_int), ("pitch", ctypes.c_int), ("y0", ctypes.c_int), ("yn", ctypes.c_int)]


class Arg(ctypes.Structure):  # mifx_cpu_arg
    _fields_ = [("kind", ctypes.c_int), ("img", Img), ("p", ctypes.c_void_p), ("bytes", ctypes.c_size_t), ("i", ctypes.c_longlong), ("f", ctypes.c_double)]


class Call(ctypes.Structure):  # mifx_cpu_call
    _fields_ = [("name", ctypes.c_char_p), ("count", ctypes.c_int), ("arg", Arg * 40), ("stream", ctypes.c_void_p)]


class Pyr(ctypes.Structure):  # mifx::Pyr
    _fields_ = [("l", Img * 8), ("levels", ctypes.c_int)]


class HizSlab(ctypes.Structure):  # mifx::HizSlab (mifx_host.h)
    _fields_ = [("base", ctypes.c_void_p), ("offset", ctypes.c_uint32 * 8), ("pitch", ctypes.c_uint32 * 8), ("w", ctypes.c_uint32 * 8), ("h", ctypes.c_uint32 * 8),
                ("levels", ctypes.c_int), ("bytes", ctypes.c_uint32), ("base0", ctypes.c_void_p), ("bytes0", ctypes.c_uint32)]


class CamK(ctypes.Structure):  # mifx::CamK (mifx_device.h)
    _fields_ = [("view", ctypes.c_float * 16), ("proj", ctypes.c_float * 16), ("viewProj", ctypes.c_float * 16), ("viewInv", ctypes.c_float * 16), ("viewProjInv", ctypes.c_float * 16),
                ("pos", ctypes.c_float * 3), ("vw", ctypes.c_float), ("vh", ctypes.c_float), ("ivw", ctypes.c_float), ("ivh", ctypes.c_float), ("jx", ctypes.c_float),
                ("jy", ctypes.c_float), ("frameIndex", ctypes.c_uint32), ("reversedDepth", ctypes.c_int)]


class SsrCleanupIn(ctypes.Structure):  # mifx::SsrCleanupIn (mifx_ssr_cleanup.h)
    _fields_ = [("depth", Img), ("roughness", Img), ("radiance", Img), ("variance", Img), ("mask", Img), ("RoughnessThreshold", ctypes.c_float),
                ("BilateralCleanupSpatialSigmaFactor", ctypes.c_float), ("AlphaInterpolation", ctypes.c_float), ("ReversedDepth", ctypes.c_int)]


TRACK = None  # tests/cpu_product/order.py LaneOrder while a run checks the order of the lanes: every view() is a read, every store() a write of the current launch


def view(img, c=1, _write=False):
    """The plane as a numpy array (h, w[, c]) of float32 over the product's own memory (rows `pitch` bytes apart)."""
    if not img.p:
        return None
    if TRACK is not None:
        TRACK.access(img, _write)
    floats = img.pitch // 4
    flat = np.ctypeslib.as_array((ctypes.c_float * (floats * img.h)).from_address(img.p))
    a = flat.reshape(img.h, floats)[:, : img.w * c]
    return a.reshape(img.h, img.w, c) if c > 1 else a


def store(img, values, c=1):
    """Writes a pass's output: the whole plane, or -- when the launcher was given a row window (Img::y0 / yn: row-band sharding) -- the rows of the window only, as the kernel
    would: what lies outside keeps its content, so a consumer that reads beyond what its producers computed shows up as a difference."""
    v = view(img, c, _write=True)
    if img.yn:
        v[img.y0: img.y0 + img.yn] = values[img.y0: img.y0 + img.yn] if np.ndim(values) else values
    else:
        v[...] = values


def tight(a):
    return np.ascontiguousarray(a, np.float32)


def blob(arg, cls):
    return cls.from_address(arg.p)


class Device:
    """Holds what the launchers do not carry: the checker, and the CameraAttribs blocks of the frame (the launchers take the kernels' own camera form, CamK; the reference's
    shaders read the full block -- the patched PostFXContext.execute of run.py notes them; every handler checks that the CamK it was given belongs to the block it uses)."""

    def __init__(self, lib, prefix="ref_"):
        self.lib, self.prefix = lib, prefix
        self.cam = self.prev_cam = None
        self._lock = threading.Lock()
        self._tls = threading.local()  # (what one depth-of-field pass leaves for the next: per thread -- the ranks of an in-process group run on threads)
        self.dof_attribs = None  # DOFAttribs bytes of the frame (noted by run.py's patched DepthOfField.execute: the launchers carry scalars of it)
        self.log = []
        self.algorithm = "gtao"
        self.taa_flags_seen = None

    def chain(self, rev):
        return cpu_chain.CpuChain(self.lib, self.prefix, reversed_depth=bool(rev))

    def camera(self, k, which="cur"):
        from diligentfx_amd import binding as B

        raw = self.cam if which == "cur" else self.prev_cam
        block = B.camera_from_bytes(raw)
        assert list(k.viewProj) == list(block.mViewProj) and list(k.proj) == list(block.mProj), f"the {which} camera of this launch is not the frame's {which} camera"
        return raw

    # ------------------------------------------------------------------------------------------------ dispatch
    def launch(self, call_ptr):
        c = call_ptr.contents
        name = c.name.decode()
        a = [c.arg[i] for i in range(c.count)]
        self.log.append(name)
        try:
            with self._lock:  # (the checker binds its textures to globals: one pass at a time, whichever rank's thread asks)
                if TRACK is not None:
                    TRACK.begin_launch(c.stream or 0, name)
                try:
                    getattr(self, "do_" + name)(*a)
                finally:
                    if TRACK is not None:
                        TRACK.end_launch()
            return 0
        except Exception as e:  # noqa: BLE001 -- reported through the library's status, with the reason on stderr
            import traceback

            traceback.print_exc()
            print(f"tests/cpu_product: launch_{name} failed: {e!r}", flush=True)
            return -5

    # ------------------------------------------------------------------------------------------------ plumbing passes
    def do_fill_f32(self, plane, floats_per_texel, value):
        view(plane.img, int(floats_per_texel.i), _write=True)[...] = np.float32(value.f)

    def do_depth16_copy(self, src, dst):
        store(dst.img, view(src.img))  # (fp32 build: a plain copy; FEATURE_FLAG_HALF_PRECISION_DEPTH quantises in the native-storage build only)

    # ------------------------------------------------------------------------------------------------ PostFXContext (C1-C3)
    def do_postfx_prep(self, depth, motion, reproj, closest, cur, prev, sobol, tile, noise_xy, noise_zw, frame, half_precision_depth):
        k = blob(cur, CamK)
        ch = self.chain(k.reversedDepth)
        cam, prev_cam = self.camera(k), self.camera(blob(prev, CamK), "prev")
        s = np.ctypeslib.as_array((ctypes.c_uint8 * 256).from_address(sobol.p)).astype(np.float32).reshape(1, 256)
        t = np.ctypeslib.as_array((ctypes.c_uint8 * (128 * 128 * 8)).from_address(tile.p)).astype(np.float32).reshape(256, 512)
        xy, zw = cpu_chain.f32((128, 128, 2)), cpu_chain.f32((128, 128, 2))
        ch.call("blue_noise", [s, t], [xy, zw], ival=[int(frame.i)])
        store(noise_xy.img, xy, 2)
        store(noise_zw.img, zw, 2)
        d = tight(view(depth.img))
        rd = cpu_chain.f32(d.shape)
        ch.call("reprojected_depth", [d], [rd], cam0=cam, cam1=prev_cam)
        store(reproj.img, rd)
        m = tight(view(motion.img, 2))
        cm = cpu_chain.f32(m.shape)
        ch.call("closest_motion", [d, m], [cm])
        store(closest.img, cm, 2)

    # ------------------------------------------------------------------------------------------------ SSAO (A2, A3, A5-A8; full resolution, plain passes)
    def do_ssao_prefilter_pyramid(self, p, camz, cam, attribs, depth16):
        pyr, k = blob(p, Pyr), blob(cam, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        src = tight(view(pyr.l[0]))
        for lv in range(1, pyr.levels):
            o = cpu_chain.f32((pyr.l[lv].h, pyr.l[lv].w))
            ch.call("ssao_prefiltered_depth_mip", [src], [o], cam0=self.camera(k), attribs=ab, ival=[lv - 1])
            store(pyr.l[lv], o)  # (row-band sharding: a level may carry a STORE window -- the rows A3's taps at that level can reach; the next level is reduced from the whole one)
            src = o
        # (the camera-z twin of the pyramid is the kernels' own acceleration structure: the reference's passes read depth)

    def do_ssao_downsample_depth(self, depth, out):  # A1 (half resolution): the checkerboard depth
        o = cpu_chain.f32((out.img.h, out.img.w))
        self.chain(False).call("ssao_downsampled_depth", [tight(view(depth.img))], [o])
        store(out.img, o)

    def do_ssao_compute_ao(self, depth_pyr, camz_pyr, normal, noise_zw, out, cam, attribs, half_resolution, half_precision_depth):
        pyr, k = blob(depth_pyr, Pyr), blob(cam, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        from diligentfx_amd import binding as B

        algo = ("gtao", "hbao", "vbao")[B.SSAOAttribs.from_buffer_copy(ab).Algorithm]
        levels = [tight(view(pyr.l[i])) for i in range(pyr.levels)]
        o = cpu_chain.f32((out.img.h, out.img.w), 1.0)
        ins = [levels, tight(view(normal.img, 4)), tight(view(noise_zw.img, 2))]
        if half_resolution.i:  # (the reference build has one entry point per permutation, for the plain depth convention)
            assert not k.reversedDepth and not half_precision_depth.i
            ch.call(f"ssao_compute_ao_{algo}_half", ins, [o], cam0=self.camera(k), attribs=ab)
        elif half_precision_depth.i:
            assert algo == "gtao" and not k.reversedDepth
            ch.call("ssao_compute_ao_gtao_halfprec", ins, [o], cam0=self.camera(k), attribs=ab)
        else:
            ch.call("ssao_compute_ao_" + algo, ins, [o], cam0=self.camera(k), attribs=ab)
        store(out.img, o)

    def do_ssao_bilateral_upsample(self, depth, occl, out, cam):  # A4 (half resolution)
        from diligentfx_amd import binding as B

        k = blob(cam, CamK)
        o = cpu_chain.f32((out.img.h, out.img.w))
        self.chain(k.reversedDepth).call("ssao_bilateral_upsampling", [tight(view(depth.img)), tight(view(occl.img))], [o], cam0=self.camera(k), attribs=bytes(B.SSAOAttribs.default()))
        store(out.img, o)

    def do_ssao_depth_to_camz(self, depth, camz, cam):
        pass  # (feeds the kernels' camera-z taps only)

    def do_ssao_temporal(self, curr_ao, prev_ao, prev_len, reproj_depth, prev_depth, motion, out_ao, out_len, cur, prev, attribs, resolve):
        # (resolve: the fused resolve -- the kernel also prepares A7 / A8 for the list pass below; the values are those of the plain passes: do_ssao_resolve_lists)
        k = blob(cur, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        h, w = out_ao.img.h, out_ao.img.w
        o_ao, o_len = cpu_chain.f32((h, w), 1.0), cpu_chain.f32((h, w), 1.0)
        ch.call("ssao_temporal_accumulation", [tight(view(curr_ao.img)), tight(view(prev_ao.img)), tight(view(prev_len.img)), tight(view(reproj_depth.img)), tight(view(prev_depth.img)),
                                               tight(view(motion.img, 2))], [o_ao, o_len], cam0=self.camera(k), cam1=self.camera(blob(prev, CamK), "prev"), attribs=ab)
        store(out_ao.img, o_ao)
        store(out_len.img, o_len)

    def do_ssao_resolve_lists(self, ao_pyr, depth_pyr, hist_len, camz, normal, rows5, resolve, cam, attribs):
        """The fused resolve (ssao.hip): A7 and A8 for the pixels that need them, from work lists the temporal kernel filled -- "the same value for every texel as the two
        full-frame passes" (api_ssao.cpp): here the two full-frame passes, on the planes the resolve names."""
        class SsaoResolve(ctypes.Structure):
            _fields_ = [("depth", Img), ("resampled", Img), ("out", Img), ("out2", Img), ("lists", ctypes.c_void_p)]

        r = SsaoResolve.from_address(resolve.p)
        ap, dp, k = blob(ao_pyr, Pyr), blob(depth_pyr, Pyr), blob(cam, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        h, w = r.out.h, r.out.w
        res = cpu_chain.f32((h, w))
        ch.call("ssao_resampled_history", [[tight(view(ap.l[i])) for i in range(ap.levels)], [tight(view(dp.l[i])) for i in range(dp.levels)], tight(view(hist_len.img)),
                                           tight(view(normal.img, 4))], [res], cam0=self.camera(k))
        store(r.resampled, res)
        o = cpu_chain.f32((h, w))
        ch.call("ssao_spatial_reconstruction", [res, tight(view(hist_len.img)), tight(view(r.depth)), tight(view(normal.img, 4))], [o], cam0=self.camera(k), attribs=ab)
        store(r.out, o)
        if r.out2.p:
            store(r.out2, o)

    def do_ssao_convolute_pyramids(self, ao, depth, depth16):
        ap, dp = blob(ao, Pyr), blob(depth, Pyr)
        ch = self.chain(False)
        for lv in range(1, ap.levels):
            o0, o1 = cpu_chain.f32((ap.l[lv].h, ap.l[lv].w)), cpu_chain.f32((dp.l[lv].h, dp.l[lv].w))
            ch.call("ssao_convoluted_history_mip", [tight(view(ap.l[lv - 1])), tight(view(dp.l[lv - 1]))], [o0, o1], ival=[lv - 1])
            store(ap.l[lv], o0)
            store(dp.l[lv], o1)

    def do_ssao_resample(self, ao_pyr, depth_pyr, hist_len, normal, out, cam):
        ap, dp, k = blob(ao_pyr, Pyr), blob(depth_pyr, Pyr), blob(cam, CamK)
        ch = self.chain(k.reversedDepth)
        o = cpu_chain.f32((out.img.h, out.img.w))
        ch.call("ssao_resampled_history", [[tight(view(ap.l[i])) for i in range(ap.levels)], [tight(view(dp.l[i])) for i in range(dp.levels)], tight(view(hist_len.img)),
                                           tight(view(normal.img, 4))], [o], cam0=self.camera(k))
        store(out.img, o)

    def do_ssao_spatial(self, occl, hist_len, depth, camz, normal, out, history_out, cam, attribs):
        k = blob(cam, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        o = cpu_chain.f32((out.img.h, out.img.w))
        ch.call("ssao_spatial_reconstruction", [tight(view(occl.img)), tight(view(hist_len.img)), tight(view(depth.img)), tight(view(normal.img, 4))], [o], cam0=self.camera(k), attribs=ab)
        store(out.img, o)
        if history_out.img.p:  # the copy of the resolved AO into the history slot (ScreenSpaceAmbientOcclusion.cpp:1319-1328), fused into A8 by the product
            store(history_out.img, o)

    # ------------------------------------------------------------------------------------------------ SSR (R1, R2, R4-R7; full resolution)
    def do_ssr_hiz_pyramid(self, p, level0_copy, reversed_depth):
        pyr = blob(p, Pyr)
        ch = self.chain(reversed_depth.i)
        if level0_copy.img.p:  # (null: the march reads level 0 from the depth plane itself -- HizSlab.base0, row-band frames since round 6)
            store(level0_copy.img, view(pyr.l[0]))  # (level 0 of the slab the march reads: the depth itself, a copy in the reference too, :789-806)
        src = tight(view(pyr.l[0]))
        for lv in range(1, pyr.levels):
            o = cpu_chain.f32((pyr.l[lv].h, pyr.l[lv].w))
            ch.call("ssr_hiz_mip", [src], [o], ival=[lv - 1])
            store(pyr.l[lv], o)
            src = o

    def do_ssr_mask_roughness(self, material, depth, roughness, mask, attribs, reversed_depth):
        ch = self.chain(reversed_depth.i)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        h, w = roughness.img.h, roughness.img.w
        r, m = cpu_chain.f32((h, w)), cpu_chain.f32((h, w))
        ch.call("ssr_mask_roughness", [tight(view(material.img, 4)), tight(view(depth.img))], [r, m], attribs=ab)
        store(roughness.img, r)
        store(mask.img, m)

    def do_ssr_downsampled_mask(self, roughness, depth, mask, attribs, reversed_depth):  # R3 (half resolution)
        o = cpu_chain.f32((mask.img.h, mask.img.w))
        self.chain(reversed_depth.i).call("ssr_downsampled_mask", [tight(view(roughness.img)), tight(view(depth.img))], [o], attribs=ctypes.string_at(attribs.p, attribs.bytes))
        store(mask.img, o)

    def do_ssr_intersection(self, radiance, normal, roughness, noise_xy, hiz, mask, motion, out_spec, out_dirpdf, cam, attribs, previous_frame, half_resolution, hit_coords, local_begin, local_end):
        # (hit_coords: row-band sharding -- the march records where each ray hit and a second pass fetches the colour, shading the pixel if this rank did not.  Here the
        #  shade handler writes the whole frame on every rank, so the colours are taken as in the unsharded pass and every hit is marked "nothing to fetch": the windows of all
        #  the OTHER passes are what a banded run of this build tests)
        if hit_coords.img.p:
            view(hit_coords.img, _write=True).view(np.uint32)[...] = 0xFFFFFFFF
        k, slab = blob(cam, CamK), blob(hiz, HizSlab)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        levels = []
        for lv in range(slab.levels):
            base = slab.base0 if (lv == 0 and slab.base0) else slab.base  # (level 0 where it lies: the caller's depth plane)
            if lv == 0 and slab.base0:
                assert slab.offset[0] == 0 and slab.bytes0 == slab.pitch[0] * slab.h[0], "HizSlab: level 0 outside the slab needs offset 0 and the plane's own extent"
            im = Img(base + slab.offset[lv], slab.w[lv], slab.h[lv], slab.pitch[lv], 0, 0)
            levels.append(tight(view(im)))
        h, w = out_spec.img.h, out_spec.img.w
        spec, dirpdf = cpu_chain.f32((h, w, 4)), cpu_chain.f32((h, w, 4))
        ins = [tight(view(radiance.img, 4)), tight(view(normal.img, 4)), tight(view(roughness.img)), tight(view(noise_xy.img, 2)), levels, tight(view(mask.img)), tight(view(motion.img, 2))]
        if half_resolution.i:
            assert self.prefix == "ref_" and not previous_frame.i and not k.reversedDepth
            ch.call("ssr_intersection_half", ins, [spec, dirpdf], cam0=self.camera(k), attribs=ab)
        elif self.prefix == "ref_":
            ch.call("ssr_intersection_prev" if previous_frame.i else "ssr_intersection", ins, [spec, dirpdf], cam0=self.camera(k), attribs=ab)
        else:
            ch.call("ssr_intersection", ins, [spec, dirpdf], cam0=self.camera(k), attribs=ab, ival=[int(previous_frame.i)])
        store(out_spec.img, spec, 4)
        store(out_dirpdf.img, dirpdf, 4)

    def do_ssr_spatial(self, roughness, normal, depth, dirpdf, spec, mask, out_rad, out_var, out_depth, cam, attribs, half_resolution):
        k = blob(cam, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        # (the targets are never cleared: outside the mask they keep their content -- the checker's pass writes under the mask only, into what the planes hold)
        rad, var, dep = tight(view(out_rad.img, 4)), tight(view(out_var.img)), tight(view(out_depth.img))
        ch.call("ssr_spatial_reconstruction_half" if half_resolution.i else "ssr_spatial_reconstruction",
                [tight(view(roughness.img)), tight(view(normal.img, 4)), tight(view(depth.img)), tight(view(dirpdf.img, 4)), tight(view(spec.img, 4)), tight(view(mask.img))],
                [rad, var, dep], cam0=self.camera(k), attribs=ab)
        store(out_rad.img, rad, 4)
        store(out_var.img, var)
        store(out_depth.img, dep)

    def do_ssr_temporal(self, motion, hit_depth, reproj_depth, curr_rad, curr_var, prev_depth, prev_rad, prev_var, mask, out_rad, out_var, cur, prev, attribs):
        k = blob(cur, CamK)
        ch = self.chain(k.reversedDepth)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        rad, var = tight(view(out_rad.img, 4)), tight(view(out_var.img))
        ch.call("ssr_temporal_accumulation", [tight(view(motion.img, 2)), tight(view(hit_depth.img)), tight(view(reproj_depth.img)), tight(view(curr_rad.img, 4)), tight(view(curr_var.img)),
                                              tight(view(prev_depth.img)), tight(view(prev_rad.img, 4)), tight(view(prev_var.img)), tight(view(mask.img))], [rad, var],
                cam0=self.camera(k), cam1=self.camera(blob(prev, CamK), "prev"), attribs=ab)
        store(out_rad.img, rad, 4)
        store(out_var.img, var)

    def do_ssr_bilateral(self, normal, cleanup, out, cam):
        from diligentfx_amd import binding as B

        k, r7 = blob(cam, CamK), blob(cleanup, SsrCleanupIn)
        ch = self.chain(k.reversedDepth)
        a = B.SSRAttribs.default()  # (the pass reads these three attributes only)
        a.RoughnessThreshold, a.BilateralCleanupSpatialSigmaFactor, a.AlphaInterpolation = r7.RoughnessThreshold, r7.BilateralCleanupSpatialSigmaFactor, r7.AlphaInterpolation
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        ch.call("ssr_bilateral_cleanup", [tight(view(r7.depth)), tight(view(normal.img, 4)), tight(view(r7.roughness)), tight(view(r7.radiance, 4)), tight(view(r7.variance)),
                                          tight(view(r7.mask))], [o], cam0=self.camera(k), attribs=bytes(a))
        store(out.img, o, 4)

    # ------------------------------------------------------------------------------------------------ TAA (T1)
    def do_taa(self, curr_color, prev_color, motion, reproj_depth, prev_depth, out, cur, prev, attribs, flags, fused):
        k = blob(cur, CamK)
        ch = self.chain(False)
        ab = ctypes.string_at(attribs.p, attribs.bytes)
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        if fused.p:
            # MIFX_CHAIN_FUSE_COMPOSITE_INTO_TAA: the kernel evaluates the composite (with R7 inside) for its colour tile instead of reading `curr_color`, which the
            # chain then never wrote -- here: the composite pass into a private image, TAA on that.  TaaFusedComposite = {const mifx_composite_attribs*, const SsrCleanupIn*}
            ptrs = (ctypes.c_void_p * 2).from_address(fused.p)
            assert ptrs[0] and ptrs[1], "tests/cpu_product: fused composite without its attribs / cleanup inputs"
            colour = self.composite_image(ptrs[0], ptrs[1], out.img.h, out.img.w)
        else:
            colour = tight(view(curr_color.img, 4))
        ch.call(f"taa_flags{int(flags.i)}", [colour, tight(view(prev_color.img, 4)), tight(view(motion.img, 2)), tight(view(reproj_depth.img)),
                                               tight(view(prev_depth.img))], [o], cam0=self.camera(k), cam1=self.camera(blob(prev, CamK), "prev"), attribs=ab)
        store(out.img, o, 4)

    # ------------------------------------------------------------------------------------------------ Bloom (B1-B3)
    def do_bloom_prefilter(self, src, out, attribs, packed_input):
        assert not packed_input.i  # (fp32 build: every plane is float)
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        self.chain(False).call("bloom_prefilter", [tight(view(src.img, 4))], [o], attribs=ctypes.string_at(attribs.p, attribs.bytes))
        store(out.img, o, 4)

    def do_bloom_downsample(self, src, out):
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        self.chain(False).call("bloom_downsample", [tight(view(src.img, 4))], [o])
        store(out.img, o, 4)

    def do_bloom_upsample(self, inp, down, out, attribs, final_pass, packed_input):
        assert not packed_input.i
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        self.chain(False).call("bloom_upsample", [tight(view(inp.img, 4)), tight(view(down.img, 4))], [o], attribs=ctypes.string_at(attribs.p, attribs.bytes), ival=[3 if final_pass.i else 0])
        store(out.img, o, 4)

    # ------------------------------------------------------------------------------------------------ depth of field (D1-D10; the launchers carry scalars of DOFAttribs, the
    # reference's passes read the block: it is noted per frame like the cameras, and every scalar a launcher does carry is checked against it)
    def _dof(self):
        from diligentfx_amd import binding as B

        return self.dof_attribs, B.DOFAttribs.from_buffer_copy(self.dof_attribs)

    def do_dof_coc(self, depth, out, cam, max_coc):
        ab, a = self._dof()
        assert np.float32(a.MaxCircleOfConfusion) == np.float32(max_coc.f)
        self._tls.dof_cam = ctypes.string_at(cam.p, cam.bytes)
        o = cpu_chain.f32((out.img.h, out.img.w))
        self.chain(False).call("dof_coc", [tight(view(depth.img))], [o], cam0=self._tls.dof_cam, attribs=ab)
        store(out.img, o)

    def do_dof_temporal_coc(self, curr, prev, motion, out, cam, stability):
        ab, a = self._dof()
        assert np.float32(a.TemporalStabilityFactor) == np.float32(stability.f)
        o = cpu_chain.f32((out.img.h, out.img.w))
        self.chain(False).call("dof_temporal_coc", [tight(view(curr.img)), tight(view(prev.img)), tight(view(motion.img, 2))], [o], cam0=ctypes.string_at(cam.p, cam.bytes), attribs=ab)
        store(out.img, o)

    def do_dof_dilation(self, coc, levels):
        ch = self.chain(False)
        src = cpu_chain.f32((coc.img.h, coc.img.w))
        ch.call("dof_separated_coc", [tight(view(coc.img))], [src])  # (D3: the product reads the near-field CoC through the signed one instead of storing it)
        lv = (Img * 3).from_address(levels.p)
        for i in range(3):
            o = cpu_chain.f32((lv[i].h, lv[i].w))
            ch.call("dof_dilation_coc", [src], [o])
            store(lv[i], o)
            src = o

    def do_dof_blur(self, src, out, weights):
        ch = self.chain(False)
        gauss = np.ctypeslib.as_array((ctypes.c_float * 13).from_address(weights.p)).astype(np.float32).reshape(1, 13).copy()
        bx, by = cpu_chain.f32((out.img.h, out.img.w)), cpu_chain.f32((out.img.h, out.img.w))
        ch.call("dof_blur_x", [tight(view(src.img)), gauss], [bx])
        ch.call("dof_blur_y", [bx, gauss], [by])
        store(out.img, by)

    def do_dof_prefilter(self, color, coc, dilation, out_near, out_far):
        ab, _ = self._dof()
        self._tls.dof_used_coc = tight(view(coc.img))  # (D10 is handed the circle of confusion as well; the product's pass takes it from the bokeh textures' alpha)
        n, f = cpu_chain.f32((out_near.img.h, out_near.img.w, 4)), cpu_chain.f32((out_far.img.h, out_far.img.w, 4))
        self.chain(False).call("dof_prefilter", [tight(view(color.img, 4)), tight(view(coc.img)), tight(view(dilation.img))], [n, f], attribs=ab)
        store(out_near.img, n, 4)
        store(out_far.img, f, 4)

    def _kernel(self, kernel, count, width):
        k = cpu_chain.f32((1, width, 2))
        k[0, :count] = np.ctypeslib.as_array((ctypes.c_float * (2 * count)).from_address(kernel.p)).reshape(count, 2)
        return k

    def do_dof_bokeh_gather(self, near, far, radiance, out_near, out_far, kernel, sample_count, max_coc, aspect, karis):
        ab, a = self._dof()
        assert int(sample_count.i) == 1 + a.BokehKernelRingDensity * (a.BokehKernelRingCount - 1) * a.BokehKernelRingCount // 2
        n, f = cpu_chain.f32((out_near.img.h, out_near.img.w, 4)), cpu_chain.f32((out_far.img.h, out_far.img.w, 4))
        self.chain(False).call("dof_bokeh_first_karis" if karis.i else "dof_bokeh_first", [tight(view(near.img, 4)), tight(view(far.img, 4)), self._kernel(kernel, int(sample_count.i), 128),
                                                                                            tight(view(radiance.img, 4))], [n, f], cam0=self._tls.dof_cam, attribs=ab)
        store(out_near.img, n, 4)
        store(out_far.img, f, 4)

    def do_dof_bokeh_fill(self, near, far, out_near, out_far, kernel, sample_count, max_coc, aspect):
        ab, _ = self._dof()
        n, f = cpu_chain.f32((out_near.img.h, out_near.img.w, 4)), cpu_chain.f32((out_far.img.h, out_far.img.w, 4))
        self.chain(False).call("dof_bokeh_second", [tight(view(near.img, 4)), tight(view(far.img, 4)), self._kernel(kernel, int(sample_count.i), 16)], [n, f], cam0=self._tls.dof_cam, attribs=ab)
        store(out_near.img, n, 4)
        store(out_far.img, f, 4)

    def do_dof_postfilter(self, near, far, out_near, out_far):
        n, f = cpu_chain.f32((out_near.img.h, out_near.img.w, 4)), cpu_chain.f32((out_far.img.h, out_far.img.w, 4))
        self.chain(False).call("dof_postfilter", [tight(view(near.img, 4)), tight(view(far.img, 4))], [n, f])
        store(out_near.img, n, 4)
        store(out_far.img, f, 4)

    def do_dof_combine(self, color, near, far, out, alpha):
        ab, a = self._dof()
        assert np.float32(a.AlphaInterpolation) == np.float32(alpha.f)
        coc = self._tls.dof_used_coc
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        self.chain(False).call("dof_combine", [tight(view(color.img, 4)), coc, tight(view(near.img, 4)), tight(view(far.img, 4))], [o], cam0=self._tls.dof_cam, attribs=ab)
        store(out.img, o, 4)

    # ------------------------------------------------------------------------------------------------ the chain's own launchers: shade, composite, Bloom's final pass + tone map
    @staticmethod
    def image(ptr_or_struct, c):
        """A mifx_image2d (by address or as a ctypes struct) as a tight numpy array."""
        from diligentfx_amd import binding as B

        im = B.Image2D.from_address(ptr_or_struct) if isinstance(ptr_or_struct, int) else ptr_or_struct
        return tight(view(Img(im.data, im.width, im.height, im.pitch_bytes, 0, 0), c)), im

    def do_pbr_shade(self, apron, g, camera, attribs, ibl, background, out_radiance, out_spec, row_begin, row_end, reversed_depth, shadows, ssr_mask):
        from diligentfx_amd import binding as B

        assert not shadows.p, "tests/cpu_product: the chain's shade without shadow maps"
        gb, ib = B.GBuffer.from_address(g.p), B.IBL.from_address(ibl.p)
        ch = self.chain(reversed_depth.i)
        img = lambda p, c: self.image(ctypes.addressof(p.contents), c)[0] if p else None  # noqa: E731
        lut, lim = self.image(ctypes.addressof(ib.brdf_lut.contents), 1)
        lc = {B.FORMAT_F32X2: 2, B.FORMAT_F32X4: 4}[lim.format]
        lut = self.image(ctypes.addressof(ib.brdf_lut.contents), lc)[0]

        def cube(cp):
            cm = cp.contents
            return [np.ctypeslib.as_array((ctypes.c_float * (6 * (cm.size >> m) * (cm.size >> m) * 4)).from_address(cm.mip_data[m])).reshape(6 * (cm.size >> m), cm.size >> m, 4).copy()
                    for m in range(cm.mip_count)]

        rad_img = B.Image2D.from_address(out_radiance.p)
        h, w = rad_img.height, rad_img.width
        rad, spec = cpu_chain.f32((h, w, 4)), cpu_chain.f32((h, w, 4))
        bg = list(np.ctypeslib.as_array((ctypes.c_float * 4).from_address(background.p))) if background.p else [0.0] * 4
        ch.call("pbr_shade", [img(gb.base_color, 4), img(gb.normal, 4), img(gb.material, 4), img(gb.depth, 1), img(gb.emissive, 4), img(gb.occlusion, 1), lut, cube(ib.irradiance),
                              cube(ib.prefiltered)], [rad, spec], cam0=ctypes.string_at(camera.p, camera.bytes), attribs=ctypes.string_at(attribs.p, attribs.bytes), fval=[float(x) for x in bg])
        view(Img(rad_img.data, w, h, rad_img.pitch_bytes, 0, 0), 4, _write=True)[...] = rad
        if out_spec.p:
            sp = B.Image2D.from_address(out_spec.p)
            view(Img(sp.data, w, h, sp.pitch_bytes, 0, 0), 4, _write=True)[...] = spec
        if ssr_mask.p:  # the by-product of the chain's shade kernel: SSR's pass R2 on the material / depth texels it reads anyway
            class SsrMaskOut(ctypes.Structure):
                _fields_ = [("roughness", Img), ("mask", Img), ("threshold", ctypes.c_float), ("perceptual", ctypes.c_int), ("channel", ctypes.c_uint), ("enabled", ctypes.c_int)]

            m = SsrMaskOut.from_address(ssr_mask.p)
            if m.enabled:
                a = B.SSRAttribs.default()
                a.RoughnessThreshold, a.IsRoughnessPerceptual, a.RoughnessChannel = m.threshold, m.perceptual, m.channel
                r, k = cpu_chain.f32((h, w)), cpu_chain.f32((h, w))
                ch.call("ssr_mask_roughness", [img(gb.material, 4), img(gb.depth, 1)], [r, k], attribs=bytes(a))
                store(m.roughness, r)
                store(m.mask, k)

    def do_composite(self, attribs, out_img, row_begin, row_end, r7):
        from diligentfx_amd import binding as B

        oi = B.Image2D.from_address(out_img.p)
        h, w = oi.height, oi.width
        o = self.composite_image(attribs.p, r7.p, h, w)
        rb, re_ = int(row_begin.i), int(row_end.i)
        store(Img(oi.data, w, h, oi.pitch_bytes, rb, re_ - rb if re_ > rb else 0), o, 4)  # (the rows of the band's window)

    def composite_image(self, attribs_p, r7_p, h, w):
        """The composite pass (M1) over the whole frame from the launcher's arguments: mifx_composite_attribs* and, when SSR's cleanup is evaluated in place, SsrCleanupIn*."""
        from diligentfx_amd import binding as B

        a = B.CompositeAttribs.from_address(attribs_p)
        img = lambda p, c: self.image(ctypes.addressof(p.contents), c)[0]  # noqa: E731
        cam = bytes(a.camera.contents)
        if r7_p:  # the fused instance: R7 evaluated in place of a load of its plane
            c = SsrCleanupIn.from_address(r7_p)
            sa = B.SSRAttribs.default()
            sa.RoughnessThreshold, sa.BilateralCleanupSpatialSigmaFactor, sa.AlphaInterpolation = c.RoughnessThreshold, c.BilateralCleanupSpatialSigmaFactor, c.AlphaInterpolation
            ssr = cpu_chain.f32((h, w, 4))
            self.chain(c.ReversedDepth).call("ssr_bilateral_cleanup", [tight(view(c.depth)), img(a.normal, 4), tight(view(c.roughness)), tight(view(c.radiance, 4)), tight(view(c.variance)),
                                                                       tight(view(c.mask))], [ssr], cam0=cam, attribs=bytes(sa))
        else:
            ssr = img(a.ssr, 4)
        lim = a.brdf_lut.contents
        lut = self.image(ctypes.addressof(lim), {B.FORMAT_F32X2: 2, B.FORMAT_F32X4: 4}[lim.format])[0]
        assert not a.tone_mapping or a.tone_mapping.contents.iToneMappingMode == 0, "tests/cpu_product: the chain composites without a tone map (TAA follows)"
        o = cpu_chain.f32((h, w, 4))
        self.chain(False).call("composite", [img(a.color, 4), img(a.specular_ibl, 4), ssr, img(a.ssao, 1), img(a.normal, 4), img(a.base_color, 4), img(a.material, 4), lut], [o], cam0=cam,
                               fval=[a.ssr_scale, a.ssao_scale])
        return o

    def do_tonemap(self, src, out, attribs, ave_log_lum, flags, ave_lum, packed_in):
        assert not ave_lum.p and not packed_in.i
        o = cpu_chain.f32((out.img.h, out.img.w, 4))
        self.chain(False).call("tonemap", [tight(view(src.img, 4))], [o], attribs=ctypes.string_at(attribs.p, attribs.bytes), fval=[ave_log_lum.f], ival=[int(flags.i)])
        store(out.img, o, 4)

    def do_bloom_final_tonemap(self, inp, down, out, ldr, bloom_attribs, tm_attribs, ave_log_lum, flags, write_bloom_output, packed_input):
        assert not packed_input.i
        ch = self.chain(False)
        o = cpu_chain.f32((ldr.img.h, ldr.img.w, 4))
        ch.call("bloom_upsample", [tight(view(inp.img, 4)), tight(view(down.img, 4))], [o], attribs=ctypes.string_at(bloom_attribs.p, bloom_attribs.bytes), ival=[3])
        if write_bloom_output.i:
            store(out.img, o, 4)
        t = cpu_chain.f32(o.shape)
        ch.call("tonemap", [o], [t], attribs=ctypes.string_at(tm_attribs.p, tm_attribs.bytes), fval=[ave_log_lum.f], ival=[int(flags.i)])
        store(Img(ldr.img.p, ldr.img.w, ldr.img.h, ldr.img.pitch, out.img.y0, out.img.yn), t, 4)  # (the launch covers the row window of `out`: bloom.hip)

    def do_pbr_shade_layers(self, apron, g, layers, camera, attribs, ibl, background, out_radiance, out_spec, row_begin, row_end, reversed_depth, shadows, hit):
        """The shade with material layers: the reference's permutation for the layer set (oracle/ref/ref_pl_*.cpp); the sets that have one compiled there."""
        from diligentfx_amd import binding as B

        assert not hit.p, "tests/cpu_product: unsharded frames only"
        gb, ib, ly = B.GBuffer.from_address(g.p), B.IBL.from_address(ibl.p), B.PBRLayers.from_address(layers.p)
        img = lambda p, c: self.image(ctypes.addressof(p.contents), c)[0] if p else None  # noqa: E731
        lim = ib.brdf_lut.contents
        lut = self.image(ctypes.addressof(lim), {B.FORMAT_F32X2: 2, B.FORMAT_F32X4: 4}[lim.format])[0]

        def cube(cp):
            cm = cp.contents
            return [np.ctypeslib.as_array((ctypes.c_float * (6 * (cm.size >> m) * (cm.size >> m) * 4)).from_address(cm.mip_data[m])).reshape(6 * (cm.size >> m), cm.size >> m, 4).copy()
                    for m in range(cm.mip_count)]

        rad_img = B.Image2D.from_address(out_radiance.p)
        h, w = rad_img.height, rad_img.width
        zero4 = cpu_chain.f32((h, w, 4))
        plane = lambda p: img(p, 4) if p else zero4  # noqa: E731
        tr = zero4.copy()
        if ly.transmission:
            tr[..., 0] = img(ly.transmission, 1)

        def table(p):
            if not p:
                return cpu_chain.f32((2, 2, 4))
            c = {B.FORMAT_F32: 1, B.FORMAT_F32X2: 2, B.FORMAT_F32X4: 4}[p.contents.format]
            t = img(p, c)
            return np.ascontiguousarray(np.repeat((t if c == 1 else t[..., 0])[..., None], 4, -1))

        perm = {1: "clearcoat", 2: "sheen", 4: "anisotropy", 8: "iridescence", 16: "transmission", 31: "all"}[ly.flags]
        ins = [img(gb.base_color, 4), img(gb.normal, 4), img(gb.material, 4), img(gb.depth, 1), img(gb.emissive, 4), img(gb.occlusion, 1), lut, cube(ib.irradiance), cube(ib.prefiltered),
               [plane(ly.clearcoat), plane(ly.clearcoat_normal), plane(ly.sheen), plane(ly.anisotropy), plane(ly.tangent), plane(ly.iridescence), tr],
               [table(ly.sheen_albedo_scaling_lut), table(ly.preintegrated_charlie)]]
        if shadows.p:
            sh = B.PBRShadows.from_address(shadows.p)
            sm = sh.shadow_map.contents
            assert sm.pitch_bytes == sm.width * 4 and sm.slice_pitch_bytes == sm.pitch_bytes * sm.height
            slices = np.ctypeslib.as_array((ctypes.c_float * (sm.slices * sm.height * sm.width)).from_address(sm.data)).reshape(sm.slices, sm.height, sm.width).copy()
            infos = np.ctypeslib.as_array((ctypes.c_float * (24 * sh.shadow_map_count)).from_address(ctypes.addressof(sh.shadow_maps.contents))).reshape(1, -1).copy()
            perm = {(31, 3): "all_shadows3", (2, 5): "sheen_shadows5"}[(ly.flags, sh.pcf_filter_size)]
            ins += [[s for s in slices], infos]
        rad, spec = cpu_chain.f32((h, w, 4)), cpu_chain.f32((h, w, 4))
        bg = list(np.ctypeslib.as_array((ctypes.c_float * 4).from_address(background.p))) if background.p else [0.0] * 4
        iv = [int(bool(ly.clearcoat_normal)), int(bool(ly.tangent)), 0, 0, 0, 0, 0, int(reversed_depth.i)]
        self.lib.call(self.prefix + "pbr_shade_layers_" + perm, ins, [rad, spec], cam0=ctypes.string_at(camera.p, camera.bytes), attribs=ctypes.string_at(attribs.p, attribs.bytes), ival=iv,
                      fval=[float(x) for x in bg] + [ly.iridescence_ior, ly.anisotropy_rotation])
        view(Img(rad_img.data, w, h, rad_img.pitch_bytes, 0, 0), 4, _write=True)[...] = rad
        if out_spec.p:
            sp = B.Image2D.from_address(out_spec.p)
            view(Img(sp.data, w, h, sp.pitch_bytes, 0, 0), 4, _write=True)[...] = spec

    def do_pbr_hit_fetch(self, *args):
        pass  # (see do_ssr_intersection)

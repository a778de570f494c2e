"""oracle/cpu_chain.py -- TEST INFRASTRUCTURE: host sequencing of the reference chain on the CPU.

Drives either checker library (prefix "ref_" = the reference's shader source compiled for the CPU, prefix "oracle_" = the
hand-written restatement) pass by pass, reproducing the host logic of the reference effect classes: mip loops, render-target
clears, ping-pong by FrameDesc.Index & 1, reset rules.  Images are numpy float32 arrays (H, W[, C]).

Host logic followed (sequencing only, no arithmetic):
  PostFXContext::Execute                PostProcess/Common/src/PostFXContext.cpp:287-338
  ScreenSpaceAmbientOcclusion::Execute  PostProcess/ScreenSpaceAmbientOcclusion/src/ScreenSpaceAmbientOcclusion.cpp:348-387, 790-1329
  ScreenSpaceReflection::Execute        PostProcess/ScreenSpaceReflection/src/ScreenSpaceReflection.cpp:300-341, 757-1104
  TemporalAntiAliasing::Execute         PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp:123-300
  Bloom::Execute                        PostProcess/Bloom/src/Bloom.cpp:152-156, 288-396
  DepthOfField::Execute                 PostProcess/DepthOfField/src/DepthOfField.cpp:96-171 (tables), 173-293 (resources), 295-332 (order), 820-1114 (bindings)
  HnPostProcessTask::Execute (order)    Hydrogent/src/Tasks/HnPostProcessTask.cpp:743-948
"""
import ctypes
import math

import numpy as np

SSAO_MIPS = 5  # SSAO_DEPTH_PREFILTERED_MAX_MIP + 1
SSR_MIPS = 7   # SSR_DEPTH_HIERARCHY_MAX_MIP + 1


def mip_dims(w, h, levels):
    return [(max(w >> k, 1), max(h >> k, 1)) for k in range(levels)]


def compute_mip_levels_count(w, h):
    """DiligentCore ComputeMipLevelsCount: floor(log2(max(w, h))) + 1."""
    return int(math.floor(math.log2(max(w, h)))) + 1


def f32(shape, fill=0.0):
    return np.full(shape, fill, np.float32)


class CpuChain:
    # passes with a reversed-depth permutation in oracle/_ref (PostFXContext::FEATURE_FLAG_REVERSED_DEPTH -> *_OPTION_INVERTED_DEPTH)
    REVERSED_REF = {"closest_motion", "ssr_hiz_mip", "ssr_mask_roughness", "ssr_intersection", "ssao_compute_ao_gtao", "ssao_temporal_accumulation",
                    "ssao_resampled_history", "ssao_spatial_reconstruction", "ssr_bilateral_cleanup"}

    def __init__(self, lib, prefix, algorithm="gtao", taa_flags=2, reversed_depth=False):
        self.lib, self.p = lib, prefix
        self.algorithm, self.taa_flags = algorithm, taa_flags
        self.reversed_depth = reversed_depth
        self.reset_history()

    def reset_history(self):
        self.ssao_last = self.ssr_last = self.taa_last = None
        self.ssao_hist = self.ssr_hist = self.taa_hist = self.dof_hist = None
        self.dof_state = None
        self.taa_techniques = set()  # TAA flag sets whose technique exists (TemporalAntiAliasing.cpp:161-171, 184): the first frame of a flag set is a plain copy

    def call(self, name, *a, **k):
        if self.reversed_depth:  # ival[7] = reversed depth for the hand-written oracle; the reference build has one entry point per permutation
            iv = list(k.get("ival", ()))
            iv += [0] * (8 - len(iv))
            iv[7] = 1
            k["ival"] = iv
            if self.p == "ref_" and name in self.REVERSED_REF:
                name += "_rev"
        return self.lib.call(self.p + name, *a, **k)

    # ------------------------------------------------------------------ PrepareResources of the effects
    # The reference's caller prepares every effect every frame, executed or not (HnPostProcessTask.cpp:671-683): targets are re-created -- histories cleared -- on a change
    # of the frame size, and of those feature flags that change a target (the rules below, each pinned by executing the class: tests/test_host_sequence_vs_ref.py).  The
    # effect methods call their own rule first; a caller that leaves an effect out for a frame calls prepare() so that the effect still sees that frame's size and flags.
    def prepare(self, w, h, ssr_flags=0, ssao_flags=0, dof_flags=None):
        self._prepare_ssr(h, w, bool(ssr_flags & 2))
        self._prepare_ssao(h, w, bool(ssao_flags & 2), bool(ssao_flags & 1))
        self._prepare_taa(h, w)
        if dof_flags is not None:
            self._prepare_dof(h, w, dof_flags)

    def _prepare_ssr(self, h, w, half_resolution):
        # (a change of HALF_RESOLUTION re-creates every target like a resize; PREVIOUS_FRAME only selects another permutation of R4 -- ScreenSpaceReflection.cpp:72-85)
        if self.ssr_hist is None or self.ssr_hist["rad"][0].shape[:2] != (h, w) or self.ssr_hist["half"] != half_resolution:
            self.ssr_hist = {"rad": [f32((h, w, 4)), f32((h, w, 4))], "var": [f32((h, w)), f32((h, w))], "half": half_resolution,  # cleared to 0 (.cpp:262-280)
                             # R5's targets: created with the others, never cleared; R5 and R6 run under the reflection mask (depth test), so a texel outside it keeps what an
                             # earlier frame left there -- and R6 / R7 read such texels beside the mask's edge (.cpp:904-932, 1001-1069)
                             "res": [f32((h, w, 4)), f32((h, w)), f32((h, w))]}

    def _prepare_ssao(self, h, w, half_resolution, half_precision_depth):
        # (a change of HALF_RESOLUTION / HALF_PRECISION_DEPTH re-creates every target like a resize, ScreenSpaceAmbientOcclusion.cpp:73-81; m_LastFrameIdx is kept either way)
        if self.ssao_hist is None or self.ssao_hist["ao"][0].shape != (h, w) or self.ssao_hist["flags"] != (half_resolution, half_precision_depth):
            self.ssao_hist = {"ao": [f32((h, w), 1.0), f32((h, w), 1.0)], "len": [f32((h, w), 1.0), f32((h, w), 1.0)],  # cleared to 1 (.cpp:304-321)
                              "flags": (half_resolution, half_precision_depth)}

    def _prepare_taa(self, h, w):
        if self.taa_hist is None or self.taa_hist[0].shape[:2] != (h, w):  # (the feature flags change nothing: TemporalAntiAliasing.cpp:84-88)
            self.taa_hist = [f32((h, w, 4)), f32((h, w, 4))]

    def _prepare_dof(self, h, w, flags):
        # every target is re-created on a change of the size or of the feature flags (DepthOfField.cpp:184-193); the two CoC history targets exist with temporal smoothing only
        if getattr(self, "dof_state", None) != (h, w, flags):
            self.dof_state = (h, w, flags)
            self.dof_hist = [f32((h, w)), f32((h, w))] if flags & self.DOF_FLAG_TEMPORAL else None  # cleared to 0 (.cpp:205-223)

    def depth_copy(self, kind, a):
        """A copy of a depth plane into a target that FEATURE_FLAG_HALF_PRECISION_DEPTH makes R16_UNORM (kind "postfx": the previous depth, PostFXContext.cpp:270, 325-337;
        "ssao": mip 0 of the prefiltered and of the convoluted pyramid, ScreenSpaceAmbientOcclusion.cpp:95-97, 857, 1131): exact unless the library emulates formats."""
        f = getattr(self.lib, "store_depth16", None)
        return f(kind, a) if f is not None else a.copy()

    # ------------------------------------------------------------------ PostFXContext
    def postfx(self, frame_index, depth, prev_depth, motion, cam, prev_cam, tables):
        sobol, tile = tables
        xy, zw = f32((128, 128, 2)), f32((128, 128, 2))
        self.call("blue_noise", [sobol.astype(np.float32).reshape(1, 256), tile.astype(np.float32).reshape(256, 512)], [xy, zw], ival=[frame_index])
        rd = f32(depth.shape)
        self.call("reprojected_depth", [depth], [rd], cam0=cam, cam1=prev_cam)
        cm = f32(motion.shape)
        self.call("closest_motion", [depth, motion], [cm])
        return {"noise_xy": xy, "noise_zw": zw, "reproj_depth": rd, "closest_motion": cm, "prev_depth": self.depth_copy("postfx", prev_depth), "cam": cam, "prev_cam": prev_cam,
                "frame": frame_index}

    # ------------------------------------------------------------------ SSAO
    def ssao(self, pf, depth, normal, attribs, keep=None, half_resolution=False, half_precision_depth=False):
        """attribs: SSAOAttribs ctypes struct (ResetAccumulation is OR-ed with the frame-continuity rule, .cpp:797-800).
        half_resolution: FEATURE_FLAG_HALF_RESOLUTION -- checkerboard depth (A1), pyramid + AO at half size, bilateral upsampling (A4) (.cpp:818-838, 857, 985-1008, 1047)."""
        h, w = depth.shape
        idx = pf["frame"]
        reset = self.ssao_last is None or idx != self.ssao_last + 1 or attribs.ResetAccumulation != 0
        self.ssao_last = idx
        a = type(attribs).from_buffer_copy(bytes(attribs))
        a.ResetAccumulation = 1 if reset else 0
        ab = bytes(a)
        self._prepare_ssao(h, w, half_resolution, half_precision_depth)
        cur, prv = idx & 1, (idx + 1) & 1
        cam = pf["cam"]
        # A1 (half resolution only): checkerboard depth
        full_dims = mip_dims(w, h, SSAO_MIPS)
        src_depth, aw, ah = depth, w, h
        if half_resolution:
            aw, ah = w // 2, h // 2
            src_depth = f32((ah, aw))
            self.call("ssao_downsampled_depth", [depth], [src_depth])
        # A2: prefiltered depth pyramid (of the checkerboard depth in half-resolution mode)
        dims = mip_dims(aw, ah, SSAO_MIPS)
        pyr = [self.depth_copy("ssao", src_depth) if half_precision_depth else src_depth.copy()]
        for k in range(1, SSAO_MIPS):
            o = f32((dims[k][1], dims[k][0]))
            self.call("ssao_prefiltered_depth_mip", [pyr[k - 1]], [o], cam0=cam, attribs=ab, ival=[k - 1])
            pyr.append(o)
        # A3
        ao = f32((ah, aw), 1.0)
        if half_resolution:
            if self.p == "ref_":  # the reference build has one entry point per (algorithm, half-resolution) permutation
                self.call(f"ssao_compute_ao_{self.algorithm}_half", [pyr, normal, pf["noise_zw"]], [ao], cam0=cam, attribs=ab)
            else:
                self.call("ssao_compute_ao_" + self.algorithm, [pyr, normal, pf["noise_zw"]], [ao], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 0, 1])
            half_ao, ao = ao, f32((h, w))
            # A4
            self.call("ssao_bilateral_upsampling", [depth, half_ao], [ao], cam0=cam, attribs=ab)
            if keep is not None:
                keep.update({"ssao_checkerboard": src_depth, "ssao_ao_half": half_ao})
        elif half_precision_depth:  # FEATURE_FLAG_HALF_PRECISION_DEPTH: the self-occlusion offset of A3 (SSAO_ComputeAmbientOcclusion.fx:145-150); the planes stay fp32 here
            assert self.algorithm == "gtao" or self.p != "ref_", "the reference build has the half-precision permutation of GTAO only"
            if self.p == "ref_":
                self.call("ssao_compute_ao_gtao_halfprec", [pyr, normal, pf["noise_zw"]], [ao], cam0=cam, attribs=ab)
            else:
                self.call("ssao_compute_ao_" + self.algorithm, [pyr, normal, pf["noise_zw"]], [ao], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 1])
        else:
            self.call("ssao_compute_ao_" + self.algorithm, [pyr, normal, pf["noise_zw"]], [ao], cam0=cam, attribs=ab)
        dims = full_dims
        # A5
        hist_ao, hist_len = f32((h, w), 1.0), f32((h, w), 1.0)
        self.call("ssao_temporal_accumulation", [ao, self.ssao_hist["ao"][prv], self.ssao_hist["len"][prv], pf["reproj_depth"], pf["prev_depth"], pf["closest_motion"]],
                  [hist_ao, hist_len], cam0=cam, cam1=pf["prev_cam"], attribs=ab)
        # A6: convoluted AO-history / depth pyramids
        ao_pyr, d_pyr = [hist_ao.copy()], [self.depth_copy("ssao", depth) if half_precision_depth else depth.copy()]
        for k in range(1, SSAO_MIPS):
            o0, o1 = f32((dims[k][1], dims[k][0])), f32((dims[k][1], dims[k][0]))
            self.call("ssao_convoluted_history_mip", [ao_pyr[k - 1], d_pyr[k - 1]], [o0, o1], ival=[k - 1])
            ao_pyr.append(o0)
            d_pyr.append(o1)
        # A7
        resampled = f32((h, w))
        self.call("ssao_resampled_history", [ao_pyr, d_pyr, hist_len, normal], [resampled], cam0=cam)
        # A8 (+ copy of the resolved AO into the current history slot, .cpp:1319-1328)
        out = f32((h, w))
        self.call("ssao_spatial_reconstruction", [resampled, hist_len, depth, normal], [out], cam0=cam, attribs=ab)
        self.ssao_hist["ao"][cur] = out.copy()
        self.ssao_hist["len"][cur] = hist_len
        if keep is not None:
            keep.update({"ssao_prefiltered_depth": pyr, "ssao_ao": ao, "ssao_hist_ao": hist_ao, "ssao_hist_len": hist_len, "ssao_ao_pyr": ao_pyr,
                         "ssao_depth_pyr": d_pyr, "ssao_resampled": resampled, "ssao_out": out})
        return out

    # ------------------------------------------------------------------ SSR
    def ssr(self, pf, color, depth, normal, material, motion, attribs, keep=None, previous_frame=False, half_resolution=False):
        """previous_frame: FEATURE_FLAG_PREVIOUS_FRAME (`color` is last frame's; ScreenSpaceReflection.cpp:474, 601-602).
        half_resolution: FEATURE_FLAG_HALF_RESOLUTION -- R3 half-size mask, rays traced at half size (R4), R5 reads the half-size ray textures (.cpp:934-961, 201-213, 988)."""
        h, w = depth.shape
        idx = pf["frame"]
        ab = bytes(attribs)
        cam = pf["cam"]
        self._prepare_ssr(h, w, half_resolution)
        cur, prv = idx & 1, (idx + 1) & 1
        dims = mip_dims(w, h, SSR_MIPS)
        hiz = [depth.copy()]
        for k in range(1, SSR_MIPS):
            o = f32((dims[k][1], dims[k][0]))
            self.call("ssr_hiz_mip", [hiz[k - 1]], [o], ival=[k - 1])
            hiz.append(o)
        rough, mask = f32((h, w)), f32((h, w))
        self.call("ssr_mask_roughness", [material, depth], [rough, mask], attribs=ab)
        if half_resolution:
            assert not previous_frame and not self.reversed_depth, "the reference build has the half-resolution permutation of the plain variant only"
            hh, hw = h // 2, w // 2
            half_mask = f32((hh, hw))
            self.call("ssr_downsampled_mask", [rough, depth], [half_mask], attribs=ab)
            spec, dirpdf = f32((hh, hw, 4)), f32((hh, hw, 4))
            r4_in = [color, normal, rough, pf["noise_xy"], hiz, half_mask, motion]
            res_rad, res_var, res_depth = self.ssr_hist["res"]
            r5_in = [rough, normal, depth, dirpdf, spec, mask]
            if self.p == "ref_":
                self.call("ssr_intersection_half", r4_in, [spec, dirpdf], cam0=cam, attribs=ab)
                self.call("ssr_spatial_reconstruction_half", r5_in, [res_rad, res_var, res_depth], cam0=cam, attribs=ab)
            else:
                self.call("ssr_intersection", r4_in, [spec, dirpdf], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 0, 1])
                self.call("ssr_spatial_reconstruction", r5_in, [res_rad, res_var, res_depth], cam0=cam, attribs=ab, ival=[0, 0, 0, 0, 0, 0, 1])
            if keep is not None:
                keep["ssr_half_mask"] = half_mask
        else:
            spec, dirpdf = f32((h, w, 4)), f32((h, w, 4))
            if self.p == "ref_":
                self.call("ssr_intersection_prev" if previous_frame else "ssr_intersection", [color, normal, rough, pf["noise_xy"], hiz, mask, motion], [spec, dirpdf], cam0=cam, attribs=ab)
            else:
                self.call("ssr_intersection", [color, normal, rough, pf["noise_xy"], hiz, mask, motion], [spec, dirpdf], cam0=cam, attribs=ab, ival=[int(previous_frame)])
            res_rad, res_var, res_depth = self.ssr_hist["res"]
            self.call("ssr_spatial_reconstruction", [rough, normal, depth, dirpdf, spec, mask], [res_rad, res_var, res_depth], cam0=cam, attribs=ab)
        h_rad, h_var = self.ssr_hist["rad"][cur], self.ssr_hist["var"][cur]  # (written under the mask: elsewhere the slot keeps what the frame before last left)
        self.call("ssr_temporal_accumulation", [motion, res_depth, pf["reproj_depth"], res_rad, res_var, pf["prev_depth"], self.ssr_hist["rad"][prv],
                                                self.ssr_hist["var"][prv], mask], [h_rad, h_var], cam0=cam, cam1=pf["prev_cam"], attribs=ab)
        out = f32((h, w, 4))
        self.call("ssr_bilateral_cleanup", [depth, normal, rough, h_rad, h_var, mask], [out], cam0=cam, attribs=ab)
        if keep is not None:
            keep.update({"ssr_hiz": hiz, "ssr_roughness": rough, "ssr_mask": mask, "ssr_spec": spec, "ssr_dirpdf": dirpdf, "ssr_res_rad": res_rad.copy(),
                         "ssr_res_var": res_var.copy(), "ssr_res_depth": res_depth.copy(), "ssr_hist_rad": h_rad.copy(), "ssr_hist_var": h_var.copy(), "ssr_out": out})
        return out

    # ------------------------------------------------------------------ TAA
    def taa(self, pf, color, attribs, keep=None):
        h, w = color.shape[:2]
        idx = pf["frame"]
        reset = self.taa_last is None or idx != self.taa_last + 1 or attribs.ResetAccumulation != 0
        self.taa_last = idx
        a = type(attribs).from_buffer_copy(bytes(attribs))
        a.ResetAccumulation = 1 if reset else 0
        self._prepare_taa(h, w)
        cur, prv = idx & 1, (idx + 1) & 1
        # m_AllPSOsReady is evaluated in PrepareResources, before Execute creates the technique of this flag set (.cpp:161-171, 184): the first frame a flag set is
        # executed with takes ComputePlaceholderTexture (:191-198, 302-311) -- the colour copied into the accumulation buffer, alpha included -- and, since
        # UpdateConstantBuffer ran (:188), counts as the previous frame of the next one.  (Pinned by executing the reference's host code: tests/test_host_sequence_vs_ref.py.)
        ready = self.taa_flags in self.taa_techniques
        self.taa_techniques.add(self.taa_flags)
        if not ready:
            out = np.ascontiguousarray(color, np.float32).copy()
            self.taa_hist[cur] = out
            if keep is not None:
                keep["taa_out"] = out
            return out
        out = f32((h, w, 4))
        self.call(f"taa_flags{self.taa_flags}", [color, self.taa_hist[prv], pf["closest_motion"], pf["reproj_depth"], pf["prev_depth"]], [out],
                  cam0=pf["cam"], cam1=pf["prev_cam"], attribs=bytes(a))
        self.taa_hist[cur] = out
        if keep is not None:
            keep["taa_out"] = out
        return out

    # ------------------------------------------------------------------ Bloom
    def bloom(self, color, attribs, keep=None):
        h, w = color.shape[:2]
        ab = bytes(attribs)
        hw, hh = w // 2, h // 2
        tex_count = compute_mip_levels_count(hw, hh)
        dims = [(max(hw >> k, 1), max(hh >> k, 1)) for k in range(tex_count)]
        mip_count = int(np.float32(attribs.Radius) * np.float32(compute_mip_levels_count(hw, hh)))  # Bloom::ComputeMipCount, Bloom.cpp:152-156
        if not 2 <= mip_count <= tex_count:  # same refusal as mifx_bloom_execute: below 2 levels the reference reads an unwritten texture
            raise ValueError(f"Bloom radius {attribs.Radius} gives {mip_count} pyramid levels of {tex_count}")
        down = [None] * tex_count
        up = [None] * tex_count
        down[0] = f32((dims[0][1], dims[0][0], 4))
        self.call("bloom_prefilter", [color], [down[0]], attribs=ab)
        for i in range(1, mip_count):
            down[i] = f32((dims[i][1], dims[i][0], 4))
            self.call("bloom_downsample", [down[i - 1]], [down[i]])
        last = mip_count - 1
        for i in range(last, 0, -1):
            up[i - 1] = f32((dims[i - 1][1], dims[i - 1][0], 4))
            src = up[i] if i != last else down[i]
            self.call("bloom_upsample", [down[i - 1], src], [up[i - 1]], attribs=ab, ival=[0])
        out = f32((h, w, 4))
        # with mip_count <= 1 the reference would read an unwritten up[0]; we require mip_count >= 2 (frames >= 8x8)
        self.call("bloom_upsample", [color, up[0]], [out], attribs=ab, ival=[3])
        if keep is not None:
            keep.update({"bloom_down": [d for d in down if d is not None], "bloom_up": [u for u in up if u is not None], "bloom_out": out})
        return out

    # ------------------------------------------------------------------ Depth of field
    DOF_FLAG_TEMPORAL, DOF_FLAG_KARIS = 1, 2  # DepthOfField::FEATURE_FLAGS, DepthOfField.hpp:59-68
    DOF_DILATION_LEVELS = 4                   # RESOURCE_IDENTIFIER_CIRCLE_OF_CONFUSION_DILATION_TEXTURE_MIP0 .. _LAST_MIP (= MIP3), DepthOfField.hpp:153-157

    def dof_tables(self, rings, density):
        """Large / small Octaweb kernels and the Gauss kernel as the constructor and UpdateConstantBuffers build them (DepthOfField.cpp:96-171, 799-809)."""
        large, small, gauss = f32((1, 128, 2)), f32((1, 1 + 5 * 3, 2)), f32((1, 13))
        self.call("dof_kernel_points", [], [large], ival=[rings, density])
        self.call("dof_kernel_points", [], [small], ival=[3, 5])   # DOF_BOKEH_KERNEL_SMALL_RING_COUNT / _DENSITY
        self.call("dof_gauss_kernel", [], [gauss], ival=[6], fval=[5.0])  # DOF_GAUSS_KERNEL_RADIUS / _SIGMA
        return large, small, gauss

    def dof(self, pf, color, depth, attribs, flags=0, keep=None, tables=None):
        """color: (H, W, 4) scene colour; attribs: DOFAttribs ctypes struct (AlphaInterpolation explicit; wall clock in the reference, .cpp:797).
        tables: (large, small, gauss) kernel textures to use instead of dof_tables() (tests/test_host_sequence_vs_ref.py hands in the ones the executed host code uploaded)."""
        h, w = depth.shape
        idx = pf["frame"]
        ab = bytes(attribs)
        cam = pf["cam"]
        temporal = bool(flags & self.DOF_FLAG_TEMPORAL)
        self._prepare_dof(h, w, flags)
        large, small, gauss = tables if tables is not None else self.dof_tables(attribs.BokehKernelRingCount, attribs.BokehKernelRingDensity)
        coc = f32((h, w))
        self.call("dof_coc", [depth], [coc], cam0=cam, attribs=ab)
        used = coc
        if temporal:
            cur, prv = idx & 1, (idx + 1) & 1
            used = f32((h, w))
            self.call("dof_temporal_coc", [coc, self.dof_hist[prv], pf["closest_motion"]], [used], cam0=cam, attribs=ab)
            self.dof_hist[cur] = used
        dil = [f32((h, w))]
        self.call("dof_separated_coc", [used], [dil[0]])
        for k in range(1, self.DOF_DILATION_LEVELS):
            o = f32((h >> k, w >> k))
            self.call("dof_dilation_coc", [dil[k - 1]], [o])
            dil.append(o)
        blur_x, blur_y = f32(dil[-1].shape), f32(dil[-1].shape)
        if self.p == "ref_":
            self.call("dof_blur_x", [dil[-1], gauss], [blur_x])
            self.call("dof_blur_y", [blur_x, gauss], [blur_y])
        else:
            self.call("dof_blur", [dil[-1], gauss], [blur_x], ival=[0])
            self.call("dof_blur", [blur_x, gauss], [blur_y], ival=[1])
        hh, hw = h // 2, w // 2
        pre = [f32((hh, hw, 4)), f32((hh, hw, 4))]
        self.call("dof_prefilter", [color, used, blur_y], pre, attribs=ab)
        bokeh = [f32((hh, hw, 4)), f32((hh, hw, 4))]
        karis = bool(flags & self.DOF_FLAG_KARIS)
        if self.p == "ref_":
            self.call("dof_bokeh_first_karis" if karis else "dof_bokeh_first", [pre[0], pre[1], large, color], bokeh, cam0=cam, attribs=ab)
        else:
            self.call("dof_bokeh_first", [pre[0], pre[1], large, color], bokeh, cam0=cam, attribs=ab, ival=[int(karis)])
        fill = [f32((hh, hw, 4)), f32((hh, hw, 4))]   # written into the prefiltered textures again (.cpp:1049-1058)
        self.call("dof_bokeh_second", [bokeh[0], bokeh[1], small], fill, cam0=cam, attribs=ab)
        post = [f32((hh, hw, 4)), f32((hh, hw, 4))]   # written into the bokeh textures again (.cpp:1073-1080)
        self.call("dof_postfilter", fill, post)
        out = f32((h, w, 4))
        self.call("dof_combine", [color, used, post[0], post[1]], [out], cam0=cam, attribs=ab)
        if keep is not None:
            keep.update({"dof_coc": coc, "dof_coc_used": used, "dof_dilation": dil, "dof_blur_x": blur_x, "dof_blur_y": blur_y, "dof_prefiltered": pre,
                         "dof_bokeh": bokeh, "dof_fill": fill, "dof_post": post, "dof_out": out, "dof_tables": (large, small, gauss)})
        return out

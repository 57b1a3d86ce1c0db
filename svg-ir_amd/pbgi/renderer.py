"""`pbgi.renderer.Renderer` of the reference, the part `GaussianModel.update_radiace` uses (scene/gaussian_model.py:469-522):
`set_proxy_from_gaussian_model` (pbgi/renderer.py:429-466), `build_bvh` (:582-594) and `render_radiance_with_sampling_SH`
(:596-615) with their names, arguments, result order / shapes / dtypes and the attributes the caller reads back
(`LBVHNode_info`, `LBVHNode_aabb`, `hemi_index_buffers`, `uv_buffers`).  The slang kernels are the HIP kernels of
svg-ir_amd/csrc/pbgi.hip; no CPU / PyTorch fallback."""
import torch

from gaussian_renderer import _native

from .bvhhelpers import GsBvh, _lib


class Renderer:
    def __init__(self):
        self.proxy_xyzs = None
        self.hemi_index_buffers = None
        self.uv_buffers = None
        self.hti_indices = None
        self.proxy_rot_mats = None
        self._bvh = None
        self.LBVHNode_info = None
        self.LBVHNode_aabb = None

    def set_proxy_from_gaussian_model(self, pc):
        """Every surfel is a proxy (the reference's opacity filter is `> 0.0` and is only used for `proxy_idx`)."""
        self.set_proxy(pc.get_xyz, pc.get_scaling, pc.get_rotation, pc.get_geo_normal, pc.get_opacity, pc.get_features)

    def set_proxy(self, xyzs, scales, rotates, normals, opacity, features):
        """The tensors the tracer reads (renderer.py:442-454): xyz [P,3], scaling [P,3], rotation [P,4] (r,x,y,z), geometric
        normals [P,3], opacity [P,1] or [P], SH features [P,16,3]."""
        self.proxy_xyzs, self.proxy_scales, self.proxy_rotates = xyzs, scales, rotates
        self.proxy_normals, self.proxy_opacity, self.proxy_features = normals, opacity, features
        self.proxy_idx = torch.nonzero(opacity.reshape(-1) > 0.0)[..., 0].long()

    def build_bvh(self):
        if self.proxy_idx.shape[0] == 0:
            return
        self._bvh = GsBvh(self.proxy_xyzs, self.proxy_scales)
        self.LBVHNode_info, self.LBVHNode_aabb = self._bvh.tensors()

    @torch.no_grad()
    def render_radiance_with_sampling_SH(self, ray_o, ray_d, cov3D_inv, sample_num=64):
        """ray_o [N,3] (one origin per row), ray_d [N,sample_num,3], cov3D_inv [P,6] ->
        (radiance [N,S,3], visibility [N,S,1], hit_indices [N,S,1] int32, uvs [N,S,2])."""
        if self._bvh is None:
            raise RuntimeError("build_bvh() first")
        dev, P = self._bvh.device, self._bvh.P
        N, S = int(ray_d.shape[0]), int(sample_num)
        if tuple(ray_d.shape) != (N, S, 3) or int(self.proxy_features.shape[1]) < 16:
            raise ValueError("ray_d must be [N, sample_num, 3] and the SH features [P, >=16, 3]")
        with torch.cuda.device(dev):
            f = lambda t: _native.f32c(t.detach(), dev)
            ro, rd = f(ray_o.reshape(N, 3)), f(ray_d)
            shs = f(self.proxy_features[:, :16, :])
            args = [f(self.proxy_xyzs), f(self.proxy_scales), f(self.proxy_rotates), f(self.proxy_normals),
                    f(self.proxy_opacity.reshape(-1)), f(cov3D_inv), shs]
            rad = _native.out_tensor((N, S, 3), torch.float32, dev)
            vis = _native.out_tensor((N, S, 1), torch.float32, dev)
            hit = _native.out_tensor((N, S, 1), torch.int32, dev)
            uvs = _native.out_tensor((N, S, 2), torch.float32, dev)
            _native.check(_lib.svgir_pbgi_trace_radiance(P, self._bvh.blob.data_ptr(), N, S, _native.ptr(ro), _native.ptr(rd),
                                                        *[_native.ptr(t) for t in args], rad.data_ptr(), vis.data_ptr(), hit.data_ptr(),
                                                        uvs.data_ptr(), _native.stream_ptr(dev)), "pbgi_trace_radiance")
        return rad, vis, hit, uvs

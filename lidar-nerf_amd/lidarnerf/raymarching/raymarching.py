"""Ray utilities / occupancy grid / ragged compositing (API of lidarnerf/raymarching/raymarching.py).

Each function allocates its outputs and calls the matching lnh_* entry point on the current stream."""
import torch
from torch.autograd import Function

from .. import _hip


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o = rays_o.contiguous().float().view(-1, 3)
    rays_d = rays_d.contiguous().float().view(-1, 3)
    aabb = aabb.contiguous().float()
    _hip.require_cuda(rays_o, rays_d, aabb)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
    _hip.call("lnh_near_far_from_aabb", rays_o.data_ptr(), rays_d.data_ptr(), aabb.data_ptr(), N, float(min_near),
              nears.data_ptr(), fars.data_ptr())
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o = rays_o.contiguous().float().view(-1, 3)
    rays_d = rays_d.contiguous().float().view(-1, 3)
    _hip.require_cuda(rays_o, rays_d)
    N = rays_o.shape[0]
    coords = torch.empty((N, 2), dtype=torch.float32, device=rays_o.device)
    _hip.call("lnh_sph_from_ray", rays_o.data_ptr(), rays_d.data_ptr(), float(radius), N, coords.data_ptr())
    return coords


def morton3D(coords):
    coords = coords.contiguous().int()
    _hip.require_cuda(coords)
    N = coords.shape[0]
    out = torch.empty(N, dtype=torch.int32, device=coords.device)
    _hip.call("lnh_morton3D", coords.data_ptr(), N, out.data_ptr())
    return out


def morton3D_invert(indices):
    indices = indices.contiguous().int()
    _hip.require_cuda(indices)
    N = indices.shape[0]
    out = torch.empty((N, 3), dtype=torch.int32, device=indices.device)
    _hip.call("lnh_morton3D_invert", indices.data_ptr(), N, out.data_ptr())
    return out


def packbits(grid, thresh, bitfield=None):
    grid = grid.contiguous().float()
    _hip.require_cuda(grid)
    C, H3 = grid.shape[0], grid.shape[-1] if grid.dim() == 2 else grid[0].numel()
    N = C * H3 // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    _hip.call("lnh_packbits", grid.data_ptr(), N, float(thresh), bitfield.data_ptr())
    return bitfield


def march_capacity(N, max_steps=1024, mean_count=-1, align=-1, force_all_rays=False):
    """Rows of the marcher's sample buffers (raymarching.py:235-245 of the reference): N * max_steps, or the running mean
    of the recent marches rounded up to `align`."""
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    return int(M)


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, sample_buffer=None):
    """Returns xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] (id, offset, count).  Same argument list and the same
    output-trimming rules as the reference wrapper (raymarching.py:171-282).  `sample_buffer` (this build's renderer): a
    float32 buffer of >= 8 * march_capacity(...) elements that the caller has ALREADY cleared (lnh_lidar_march_prologue
    clears it together with the counter in its one launch) — the three sample arrays become views of it."""
    rays_o = rays_o.contiguous().float().view(-1, 3)
    rays_d = rays_d.contiguous().float().view(-1, 3)
    density_bitfield = density_bitfield.contiguous()
    _hip.require_cuda(rays_o, rays_d, density_bitfield, nears, fars)
    N = rays_o.shape[0]
    M = march_capacity(N, max_steps, mean_count, align, force_all_rays)
    dev = rays_o.device
    if sample_buffer is not None:
        if not (sample_buffer.is_cuda and sample_buffer.dtype == torch.float32 and sample_buffer.is_contiguous()
                and sample_buffer.numel() >= M * 8):
            raise RuntimeError("march_rays_train: sample_buffer must be a contiguous float32 GPU buffer of >= 8 * M elements")
        buf = sample_buffer
    else:
        # (one zero fill for the three sample buffers: the step is launch-bound, every tiny kernel costs ~5 us of GPU time)
        buf = torch.zeros(M * 8, dtype=torch.float32, device=dev)
    xyzs, dirs, deltas = buf[:M * 3].view(M, 3), buf[M * 3:M * 6].view(M, 3), buf[M * 6:M * 8].view(M, 2)
    rays = torch.empty((N, 3), dtype=torch.int32, device=dev)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
    noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
    _hip.call("lnh_march_rays_train", rays_o.data_ptr(), rays_d.data_ptr(), density_bitfield.data_ptr(), float(bound),
              float(dt_gamma), int(max_steps), N, int(C), int(H), M, nears.contiguous().data_ptr(),
              fars.contiguous().data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), rays.data_ptr(),
              step_counter.data_ptr(), noises.data_ptr())
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0].item())  # host sync, as in the reference (raymarching.py:268-275)
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


class _CompositeRaysTrain(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs = sigmas.contiguous().float(), rgbs.contiguous().float()
        deltas, rays = deltas.contiguous().float(), rays.contiguous().int()
        M, N = sigmas.shape[0], rays.shape[0]
        ws = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        depth = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        image = torch.empty((N, 3), dtype=torch.float32, device=sigmas.device)
        _hip.call("lnh_composite_rays_train_forward", sigmas.data_ptr(), rgbs.data_ptr(), deltas.data_ptr(),
                  rays.data_ptr(), M, N, float(T_thresh), ws.data_ptr(), depth.data_ptr(), image.data_ptr())
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, ws, image)
        ctx.dims = (M, N, T_thresh)
        return ws, depth, image

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):  # no depth gradient, as in the reference (raymarching.py:330)
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        gs = torch.zeros_like(sigmas)
        gc = torch.zeros_like(rgbs)
        _hip.call("lnh_composite_rays_train_backward", g_ws.contiguous().data_ptr(), g_image.contiguous().data_ptr(),
                  sigmas.data_ptr(), rgbs.data_ptr(), deltas.data_ptr(), rays.data_ptr(), ws.data_ptr(),
                  image.data_ptr(), M, N, float(T_thresh), gs.data_ptr(), gc.data_ptr())
        return gs, gc, None, None, None


composite_rays_train = _CompositeRaysTrain.apply


class _CompositeRaysTrainLidar(Function):
    """LiDAR variant of composite_rays_train: K channels (ray-drop, intensity) and the ABSOLUTE depth sum(w * z) with
    its gradient — what renderer.py:233-271 computes on dense [N,T] tensors, on the marcher's ragged samples.
    (sigmas [M], feats [M,K], deltas [M,2], xyzs [M,3], rays_o/rays_d [N,3], rays [N,3]) -> (ws [N], depth [N], image [N,K])."""

    @staticmethod
    def forward(ctx, sigmas, feats, deltas, xyzs, rays_o, rays_d, rays, T_thresh=1e-4):
        sigmas, feats = sigmas.contiguous().float(), feats.contiguous().float()
        deltas, xyzs = deltas.contiguous().float(), xyzs.contiguous().float()
        rays_o, rays_d, rays = rays_o.contiguous().float(), rays_d.contiguous().float(), rays.contiguous().int()
        _hip.require_cuda(sigmas, feats, deltas, xyzs, rays_o, rays_d, rays)
        M, N, K = sigmas.shape[0], rays.shape[0], feats.shape[-1]
        ws = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        depth = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        image = torch.empty((N, K), dtype=torch.float32, device=sigmas.device)
        _hip.call("lnh_lidar_composite_rays_train_forward", sigmas.data_ptr(), feats.data_ptr(), deltas.data_ptr(),
                  xyzs.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), rays.data_ptr(), M, N, K, float(T_thresh),
                  ws.data_ptr(), depth.data_ptr(), image.data_ptr())
        ctx.save_for_backward(sigmas, feats, deltas, xyzs, rays_o, rays_d, rays, ws, depth, image)
        ctx.dims = (M, N, K, T_thresh)
        return ws, depth, image

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):
        sigmas, feats, deltas, xyzs, rays_o, rays_d, rays, ws, depth, image = ctx.saved_tensors
        M, N, K, T_thresh = ctx.dims
        gs = torch.zeros_like(sigmas)
        gf = torch.zeros_like(feats)
        _hip.call("lnh_lidar_composite_rays_train_backward", g_ws.contiguous().float().data_ptr(),
                  g_depth.contiguous().float().data_ptr(), g_image.contiguous().float().data_ptr(), sigmas.data_ptr(),
                  feats.data_ptr(), deltas.data_ptr(), xyzs.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(),
                  rays.data_ptr(), ws.data_ptr(), depth.data_ptr(), image.data_ptr(), M, N, K, float(T_thresh),
                  gs.data_ptr(), gf.data_ptr())
        return gs, gf, None, None, None, None, None, None


composite_rays_train_lidar = _CompositeRaysTrainLidar.apply


# ----------------------------------------
# infer functions (raymarching.py:362-512)
# ----------------------------------------
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024):
    """March the first n_alive rays of rays_alive for at most n_step occupied samples from their current rays_t.
    Returns xyzs [M,3], dirs [M,3], deltas [M,2] with M = n_alive*n_step (padded to a multiple of `align`); unused
    slots are zero (delta == 0 ends a ray in composite_rays)."""
    rays_o = rays_o.contiguous().float().view(-1, 3)
    rays_d = rays_d.contiguous().float().view(-1, 3)
    _hip.require_cuda(rays_o, rays_d, density_bitfield, near, far, rays_alive, rays_t)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    dev = rays_o.device
    xyzs = torch.zeros((M, 3), dtype=torch.float32, device=dev)
    dirs = torch.zeros((M, 3), dtype=torch.float32, device=dev)
    deltas = torch.zeros((M, 2), dtype=torch.float32, device=dev)
    noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb \
        else torch.zeros(n_alive, dtype=torch.float32, device=dev)
    _hip.call("lnh_march_rays", int(n_alive), int(n_step), rays_alive.data_ptr(), rays_t.data_ptr(), rays_o.data_ptr(),
              rays_d.data_ptr(), float(bound), float(dt_gamma), int(max_steps), int(C), int(H),
              density_bitfield.contiguous().data_ptr(), near.contiguous().data_ptr(), far.contiguous().data_ptr(),
              xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), noises.data_ptr())
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """In-place accumulation of one marching round into weights_sum [N], depth [N], image [N,3]; rays that ended get
    rays_alive[n] = -1, the others their advanced rays_t."""
    sigmas, rgbs, deltas = sigmas.contiguous().float(), rgbs.contiguous().float(), deltas.contiguous().float()
    _hip.require_cuda(sigmas, rgbs, deltas, rays_alive, rays_t, weights_sum, depth, image)
    for t in (weights_sum, depth, image, rays_t):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("composite_rays: in-place outputs must be contiguous float32 tensors")
    _hip.call("lnh_composite_rays", int(n_alive), int(n_step), float(T_thresh), rays_alive.data_ptr(), rays_t.data_ptr(),
              sigmas.data_ptr(), rgbs.data_ptr(), deltas.data_ptr(), weights_sum.data_ptr(), depth.data_ptr(),
              image.data_ptr())
    return tuple()

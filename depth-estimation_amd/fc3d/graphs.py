"""HIP-graph replay of the image-formation step and of the capture simulator.

One capture (codePattern -> warp_p2c forward, backward to pattern[...,0]) is six short kernel
launches; issued one by one from Python (ctypes + autograd) they cost 125-250 us of host time per
step, which is as much as the kernels themselves take at 32x512x512 and far more at small batches.  Capturing
the step once into a HIP graph (torch.cuda.CUDAGraph: hipStreamBeginCapture / hipGraphLaunch on
ROCm) and replaying it removes that cost: the launches are the same, in the same order, on the
same buffers; only the host-side issue changes.

The library needs nothing special for this: every entry point is asynchronous on the stream it is
given, allocates nothing and returns no device-dependent value to the host.
"""
from __future__ import annotations

import torch

import geometry_pytorch as G
import optics_pytorch as O


class GraphedImageFormation:
    """Static-buffer replay of  Ip = codePattern(pattern, z_p);  Ic, grid_p2c, grid_c2p =
    warp_p2c(Ip, coord_p, coord_c, pose);  Ic.backward(upstream_grad).

    The tensors passed in are the graph's input buffers: write new data into them in place
    (``t.copy_(new)``) between replays.  ``pattern`` must be a leaf that requires grad; its
    ``.grad`` is the static output gradient, rewritten by every replay.  ``outputs`` holds the
    forward results (also static)."""

    def __init__(self, pattern, z_p, coord_p, coord_c, pose, upstream_grad, n_layers, is_conv_psf=True,
                 warmup: int = 3, unroll: int = 1, overlap: bool = False, points_from_depth: bool = False):
        """unroll > 1 additionally captures that many consecutive steps as ONE graph (same launches on
        the same static buffers, back to back), for callers that run many steps on unchanged inputs
        (benchmarks): `replay_steps(k)` then issues k // unroll launches of it plus k % unroll of the
        single step, which takes the pause between two graph launches out of all but a few steps.
        overlap=True captures warp_p2c's projection + splat launch as a second branch beside codePattern.
        points_from_depth=True: coord_p / coord_c that carry the note of points_in_camera_coords are captured in
        their depth-plane form (the kernels read the depth maps the points were made from).  The graph's inputs
        are then those depth tensors, not the point tensors: refresh THEM in place between replays; writing new
        points into coord_p / coord_c has no effect on such a graph."""
        if not (pattern.is_leaf and pattern.requires_grad):
            raise ValueError("pattern must be a leaf tensor that requires grad")
        self.pattern, self.z_p, self.coord_p, self.coord_c, self.pose = pattern, z_p, coord_p, coord_c, pose
        self.upstream_grad, self.n_layers, self.conv = upstream_grad, n_layers, is_conv_psf
        self.outputs = None
        self.graph = None
        self.unroll = max(1, int(unroll))
        self.graph_n = None
        # overlap: warp_p2c's first launch runs beside codePattern on a second stream (see _step)
        self.fork = torch.cuda.Stream(device=pattern.device) if overlap else None
        from fc3d import ops as _ops
        old = _ops.allow_points_from_depth_in_capture(bool(points_from_depth))
        try:
            self._capture(warmup)
        finally:
            _ops.allow_points_from_depth_in_capture(old)

    def _step(self):
        if self.fork is None:
            Ip = O.codePattern(self.pattern, self.z_p, is_conv_psf=self.conv, n_layers=self.n_layers)
            Ic, gp2c, gc2p = G.warp_p2c(Ip, self.coord_p, self.coord_c, self.pose)
        else:
            # the forward warp of coord_p (projection + visibility splat, bound by memory) needs nothing from
            # codePattern (bound by vector instructions): it goes to a second stream -- a second branch of
            # the graph -- and the two share the chip
            main = torch.cuda.current_stream()
            self.fork.wait_stream(main)
            with torch.cuda.stream(self.fork):
                begun = G.warp_p2c_begin(self.coord_p, self.pose)
            Ip = O.codePattern(self.pattern, self.z_p, is_conv_psf=self.conv, n_layers=self.n_layers)
            main.wait_stream(self.fork)
            Ic, gp2c, gc2p = G.warp_p2c_finish(Ip, self.coord_c, begun)
        Ic.backward(self.upstream_grad)
        return Ip, Ic, gp2c, gc2p

    def _capture(self, warmup):
        dev = self.pattern.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):              # warm-up off the default stream, as capture requires
            for _ in range(max(1, warmup)):
                self.pattern.grad = None
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.pattern.grad = None                   # so that .grad is allocated from the graph's pool
        g = torch.cuda.CUDAGraph()
        # thread_local: calls made meanwhile by other threads of the process (e.g. the collective
        # library's watchdog) do not invalidate the capture
        # captured on the stream the warm-up ran on: per-stream state the operators keep (the
        # self-clearing visibility mask of warp_p2c) is then in place and becomes part of the graph
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            outs = self._step()
        self.graph = g
        self.outputs = tuple(o.detach() for o in outs)
        self.grad = self.pattern.grad
        if self.unroll > 1:
            # the same step `unroll` times in a row.  Every step allocates its results anew from the
            # graph's pool (blocks freed by one step are reused by the next), so this graph has static
            # results of its own: those of its last step.
            gn = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gn, stream=side, capture_error_mode="thread_local"):
                for _ in range(self.unroll):
                    self.pattern.grad = None
                    outs_n = self._step()
            self.graph_n = gn
            self.outputs_n = tuple(o.detach() for o in outs_n)
            self.grad_n = self.pattern.grad
            self.pattern.grad = self.grad
        # the first launch of an instantiated graph does one-time work (its kernel arguments go to the
        # device); do it here, not in the caller's first replay
        with torch.cuda.stream(side):
            g.replay()
            if self.graph_n is not None:
                self.graph_n.replay()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

    def replay(self):
        """Runs the captured step on the current stream; returns (Ic, pattern.grad) static tensors."""
        self.graph.replay()
        self.pattern.grad = self.grad
        return self.outputs[1], self.grad

    def replay_steps(self, k: int):
        """k consecutive steps on the current inputs; returns the static (Ic, pattern.grad) of the last."""
        if k <= 0:
            raise ValueError("k must be positive")
        n = k // self.unroll if self.graph_n is not None else 0
        for _ in range(n):
            self.graph_n.replay()
        rest = k - n * self.unroll
        for _ in range(rest):
            self.graph.replay()
        if rest == 0:
            self.pattern.grad = self.grad_n
            return self.outputs_n[1], self.grad_n
        self.pattern.grad = self.grad
        return self.outputs[1], self.grad


class GraphedCapture:
    """Static-buffer replay of the capture simulator, fc3d.capture.forward_model (forward only):
    shading -> codePattern -> warp from the depth maps -> sensor model, eight launches.  At small
    batches those launches cost more host time than the kernels take; replaying them as one graph
    removes that.

    z_p, z_c, pose and the three noise tensors are the graph's input buffers: refill them in place
    (``t.copy_(new)``) between replays.  ``replay()`` returns forward_model's tuple; its tensors are
    static too and are overwritten by the next replay."""

    def __init__(self, z_p, z_c, pose, noise, n_layers=21, is_conv_psf=True, crop=None, pattern=None,
                 warmup: int = 3, intrinsics=None):
        """intrinsics: as for fc3d.capture.forward_model -- None (the reference's matrix, 800x1200 images only),
        a (cx, cy) principal point, or a [4,4] / [B,4,4] tensor (made before the capture: an upload inside the
        captured region cannot be replayed)."""
        from . import capture
        self.z_p, self.z_c, self.pose, self.noise = z_p, z_c, pose, noise
        if intrinsics is not None and z_p.shape[0] > 0:
            intrinsics = capture.resolve_intrinsics(intrinsics, z_p.shape[0], z_p.shape[1], z_p.shape[2], z_p.device)
        self._call = lambda: capture.forward_model(z_p, z_c, pose, n_layers, is_conv_psf, crop, noise, pattern,
                                                   fused=True, intrinsics=intrinsics)
        dev = z_p.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                self._call()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # on the warm-up's stream, so that the self-clearing visibility mask made there is the one captured
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"), torch.no_grad():
            self.outputs = self._call()
        with torch.cuda.stream(side):
            self.graph.replay()                    # the first launch of an instantiated graph does one-time work
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

    def replay(self):
        self.graph.replay()
        return self.outputs

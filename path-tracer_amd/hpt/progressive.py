"""Progressive / resumable accumulation: a checkpoint is the accumulator plus the sample cursor.
Samples are keyed by (pixel, sample index) and the accumulator is integer (include/hpt.h hpt_accum), so rendering
samples [0, a) now and [a, b) after a restart gives the same sums as one call over [0, b) -- bit for bit."""
import numpy as np


class ProgressiveRender:
    def __init__(self, scene, max_depth=8, seed=1, spp_per_pass=0):
        import torch
        self.torch = torch
        self.scene, self.max_depth, self.seed, self.spp_per_pass = scene, max_depth, seed, spp_per_pass
        self.npix = scene.width * scene.height
        # one stream for the torch ops on the accumulator AND the library's kernels (a NULL stream handle would send
        # the library back to its own stream, unordered against torch's)
        self.stream = torch.cuda.Stream()
        scene.ctx.set_stream(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            self.accum = torch.zeros((self.npix, 4), dtype=torch.int64, device="cuda")
            self.out = torch.empty((self.npix, 3), dtype=torch.float32, device="cuda")
        self.cursor = 0          # samples per pixel accumulated so far

    def advance(self, spp):
        """adds samples cursor+1 ..= cursor+spp of every pixel"""
        self.scene.render_accum_device(self.accum.data_ptr(), 0, 0, self.npix, spp=spp, first_sample=self.cursor,
                                       max_depth=self.max_depth, seed=self.seed, spp_per_pass=self.spp_per_pass)
        self.cursor += spp
        return self.cursor

    def image(self):
        """main.rs:110: sum / num_samples, as float HxWx3 on the host"""
        if self.cursor == 0:
            raise ValueError("no samples yet")
        with self.torch.cuda.stream(self.stream):
            self.scene.accum_resolve_device(self.accum.data_ptr(), self.out.data_ptr(), num_samples=self.cursor)
            img = self.out.view(self.scene.height, self.scene.width, 3).cpu()
        self.stream.synchronize()
        return img.numpy()

    def close(self):
        """hand the context back to its own stream (this object's stream dies with it)"""
        self.stream.synchronize()
        self.scene.ctx.set_stream(0)

    def save(self, path):
        self.stream.synchronize()
        np.savez(path, accum=self.accum.cpu().numpy(), cursor=self.cursor, seed=self.seed, max_depth=self.max_depth,
                 width=self.scene.width, height=self.scene.height)

    def load(self, path):
        z = np.load(path)     # allow_pickle stays False: plain arrays only
        if (int(z["width"]), int(z["height"])) != (self.scene.width, self.scene.height):
            raise ValueError("checkpoint is for a %dx%d image" % (int(z["width"]), int(z["height"])))
        if int(z["seed"]) != self.seed or int(z["max_depth"]) != self.max_depth:
            raise ValueError("checkpoint was rendered with another seed / depth")
        with self.torch.cuda.stream(self.stream):
            self.accum.copy_(self.torch.from_numpy(z["accum"]))
        self.stream.synchronize()
        self.cursor = int(z["cursor"])
        return self.cursor

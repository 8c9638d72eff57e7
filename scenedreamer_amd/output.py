"""Frame output off the critical path (the reference writes each PNG synchronously inside the render loop,
imaginaire/generators/scenedreamer.py:512-518, :629-631 -- that alone would cap the frame rate).

`FrameWriter.submit(img)` converts the tanh-range image to uint8 on the GPU with the reference's arithmetic
(`((img*0.5+0.5)*255).astype(uint8)`, i.e. truncation), copies it to a pinned host buffer on a side stream and hands
it to worker threads that encode and write the files; the render stream never waits for PCIe, zlib or JPEG.
PNG files are encoded by a small pool (zlib releases the GIL; one thread tops out near 40 frames/s at 960x540); the
video (`video_path`, the reference's `output_dir + '.mp4'` at 10 fps, scenedreamer.py:560, :631) is appended in frame
order by its own thread through scenedreamer_amd/mp4.py."""
import os
import queue
import threading

import numpy as np
import torch

try:
    from PIL import Image
except ImportError:  # PNG encoding needs Pillow; raw .npy frames are written otherwise
    Image = None


def to_uint8_hwc(img):
    """[1,3,H,W] in [-1,1] -> uint8 [H,W,3] RGB, same rounding as the reference's write_img (scenedreamer.py:513)."""
    x = ((img * 0.5 + 0.5) * 255).clamp_(0, 255).to(torch.uint8)
    return x[0].permute(1, 2, 0).contiguous()


class FrameWriter:
    def __init__(self, output_dir, fmt="png", png_compress_level=4, depth=8, video_path=None, fps=10, png_threads=3):
        self.dir = output_dir
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
        self.fmt = fmt if (fmt != "png" or Image is not None) else "npy"
        self.level = png_compress_level
        self.q = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.err = None
        self.frames_done = 0
        self._lock = threading.Lock()
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, png_threads if output_dir else 1))]
        self.video = None
        if video_path:
            from .mp4 import Mp4MjpegWriter
            self.video = Mp4MjpegWriter(video_path, fps=fps)
            self.vq = queue.Queue(maxsize=depth)
            self.vt = threading.Thread(target=self._run_video, daemon=True)
            self.vt.start()
        for t in self.threads:
            t.start()

    def submit(self, img, index):
        """Enqueue frame `index`; returns immediately (blocks only when `depth` frames are already in flight)."""
        if self.err:
            raise self.err
        if img.is_cuda:
            u8 = to_uint8_hwc(img)
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            ready = torch.cuda.Event()
            self.stream.wait_stream(torch.cuda.current_stream(img.device))
            with torch.cuda.stream(self.stream):
                host.copy_(u8, non_blocking=True)
                ready.record()
            u8.record_stream(self.stream)
        else:
            host, ready = to_uint8_hwc(img), None
        if self.video is not None:
            self.vq.put((index, host, ready))
        self.q.put((index, host, ready))

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            index, host, ready = item
            try:
                if ready is not None:
                    ready.synchronize()
                arr = host.numpy()
                if self.dir:
                    path = os.path.join(self.dir, f"{index:05d}.{self.fmt}")
                    if self.fmt == "png":
                        Image.fromarray(arr, "RGB").save(path, compress_level=self.level)
                    else:
                        np.save(path, arr)
                with self._lock:
                    self.frames_done += 1
            except Exception as e:  # surfaced on the next submit()/close()
                self.err = e
            finally:
                self.q.task_done()

    def _run_video(self):
        """Frames are appended in index order (submit() is called in order; the reorder buffer only guards callers that
        submit out of order)."""
        pending, nxt = {}, None
        while True:
            item = self.vq.get()
            if item is None:
                for k in sorted(pending):
                    self.video.append(pending[k])
                return
            index, host, ready = item
            try:
                if ready is not None:
                    ready.synchronize()
                if nxt is None:
                    nxt = index
                pending[index] = host.numpy()
                while nxt in pending:
                    self.video.append(pending.pop(nxt))
                    nxt += 1
            except Exception as e:
                self.err = e

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.video is not None:
            self.vq.put(None)
            self.vt.join()
            self.video.close()
        if self.err:
            raise self.err

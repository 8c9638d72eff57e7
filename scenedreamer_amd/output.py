"""Frame output off the critical path (the reference writes each PNG synchronously inside the render loop,
imaginaire/generators/scenedreamer.py:512-518, :629-631 -- that alone would cap the frame rate).

`FrameWriter.submit(img)` converts the tanh-range image to uint8 on the GPU with the reference's arithmetic
(`((img*0.5+0.5)*255).astype(uint8)`, i.e. truncation), copies it to a pinned host buffer on a side stream and hands
it to worker threads that encode and write the files; the render stream never waits for PCIe, zlib or JPEG.
PNG files are encoded by a small pool (zlib releases the GIL; one thread tops out near 40 frames/s at 960x540); the
video (`video_path`, the reference's `output_dir + '.mp4'` at 10 fps, scenedreamer.py:560, :631) is appended in frame
order by its own thread: through imageio's writer (H.264 via ffmpeg, what the reference produces) where imageio is
importable, else through scenedreamer_amd/mp4.py (Motion-JPEG in an MP4 container, no external dependency).
`write_scene_maps` writes the two per-trajectory maps of scenedreamer.py:562-563."""
import os
import queue
import threading

import numpy as np
import torch

try:
    from PIL import Image
except ImportError:  # PNG encoding needs Pillow; raw .npy frames are written otherwise
    Image = None


# Colours of the 11 biome / semantic classes in semantic_map.png (data table of scenedreamer.py:532-544, RGB 0..255)
BIOME_COLORS = ((255, 255, 178), (184, 200, 98), (188, 161, 53), (190, 255, 242), (106, 144, 38), (33, 77, 41), (86, 179, 106),
                (34, 61, 53), (35, 114, 94), (0, 0, 255), (0, 255, 0))


def write_scene_maps(output_dir, scene, division="reciprocal"):
    """semantic_map.png (argmax class of current_semantic_map [1,11,S,S] through BIOME_COLORS, RGB) and height_map.png
    (current_height_map [1,1,S,S] as write_img maps it: ((h * 0.5 + 0.5) * 255) truncated to uint8, one channel) --
    scenedreamer.py:545, :562-563.  Returns the two arrays.
    division: the reference pushes the colours through a float round trip (`colours / 255 * 2 - 1` at :544, then write_img's
    `(x * 0.5 + 0.5) * 255` truncated).  `/ 255` on its CUDA tensor is a multiplication by the float32 reciprocal in PyTorch
    ("reciprocal", default: what the reference writes when it runs on a GPU), on a CPU tensor an IEEE division ("ieee"); three
    of the 33 colour values come out one level apart."""
    os.makedirs(output_dir, exist_ok=True)
    sem = torch.argmax(torch.as_tensor(scene.current_semantic_map), dim=1)[0].cpu()
    c = np.asarray(BIOME_COLORS, np.float32)
    c = c * (np.float32(1.0) / np.float32(255.0)) if division == "reciprocal" else c / np.float32(255.0)
    table = torch.from_numpy(c * np.float32(2) - np.float32(1))
    sem_rgb = ((table[sem] * 0.5 + 0.5) * 255).numpy().astype(np.uint8)
    h = torch.as_tensor(scene.current_height_map)[0, 0].cpu().to(torch.float32)
    height = ((h * 0.5 + 0.5) * 255).numpy().astype(np.uint8)
    if Image is not None:
        Image.fromarray(sem_rgb, "RGB").save(os.path.join(output_dir, "semantic_map.png"), compress_level=4)
        Image.fromarray(height, "L").save(os.path.join(output_dir, "height_map.png"), compress_level=4)
    else:
        np.save(os.path.join(output_dir, "semantic_map.npy"), sem_rgb)
        np.save(os.path.join(output_dir, "height_map.npy"), height)
    return sem_rgb, height


class _ImageioVideo:
    """imageio.get_writer(path, fps) with the append / close interface of mp4.Mp4MjpegWriter (scenedreamer.py:560, :631-632)."""

    def __init__(self, path, fps):
        import imageio
        self.w = imageio.get_writer(path, fps=fps)

    def append(self, rgb):
        self.w.append_data(rgb)

    def close(self):
        self.w.close()


def open_video(path, fps=10, backend="auto"):
    """backend: "imageio" (H.264 through ffmpeg, the reference's writer), "mjpeg" (scenedreamer_amd/mp4.py), or "auto" =
    imageio if it is importable, else mjpeg."""
    if backend in ("auto", "imageio"):
        try:
            return _ImageioVideo(path, fps)
        except Exception:   # noqa: BLE001 -- imageio absent (ImportError), an inert stand-in (AttributeError / NotImplementedError),
            # or present without a video plugin: imageio v3 raises ValueError ("Could not find a backend ...") for .mp4 then
            if backend == "imageio":
                raise
    from .mp4 import Mp4MjpegWriter
    return Mp4MjpegWriter(path, fps=fps)


def to_uint8_hwc(img):
    """[1,3,H,W] in [-1,1] -> uint8 [H,W,3] RGB, same rounding as the reference's write_img (scenedreamer.py:513)."""
    x = ((img * 0.5 + 0.5) * 255).clamp_(0, 255).to(torch.uint8)
    return x[0].permute(1, 2, 0).contiguous()


class FrameWriter:
    def __init__(self, output_dir, fmt="png", png_compress_level=4, depth=8, video_path=None, fps=10, png_threads=3,
                 video_backend="auto"):
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
            self.video = open_video(video_path, fps, video_backend)
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
        """Frames are appended in ascending index order.  Callers normally submit consecutive indices; non-consecutive ones
        (a sharded rank submitting global frame ids f, f + N, ...) or out-of-order ones are handled by a BOUNDED reorder
        buffer: once more than `depth` frames wait for a missing index, the smallest pending index is written and the
        sequence continues from it -- host frames are never held without bound and encoding is never deferred to close()."""
        pending, nxt = {}, None
        limit = max(1, self.vq.maxsize)
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
                while pending:
                    if nxt not in pending:
                        if len(pending) <= limit and min(pending) > nxt:
                            break                       # still plausible that `nxt` arrives
                        nxt = min(pending)              # gap (strided / late indices): continue from the smallest one held
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

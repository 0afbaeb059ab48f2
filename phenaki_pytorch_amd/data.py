"""Host-side image / video I/O and datasets with the reference's names and call signatures (`/root/reference/phenaki_pytorch/data.py:48-265`:
ImageDataset, VideoDataset, video_tensor_to_gif, gif_to_tensor, video_to_tensor, tensor_to_video, DataLoader), built on PIL + numpy + torch
only -- this build depends on neither torchvision nor OpenCV.  Tensors are f32 in [0, 1], images (c, h, w), videos (c, f, h, w).  MP4 files go
through OpenCV when `cv2` is importable (as in the reference) and fail with a clear message when it is not.  SURVEY.md 8f row 4.
"""
import random
from pathlib import Path

import numpy as np
import torch
from PIL import Image
from torch.utils import data as torch_data

_PIL_MODE = {1: 'L', 3: 'RGB', 4: 'RGBA'}
CHANNELS_TO_MODE = _PIL_MODE                      # (the reference's public name)


def exists(val):
    return val is not None


def identity(t, *args, **kwargs):
    return t


def pair(val):
    return val if isinstance(val, tuple) else (val, val)


# ------------------------------------------------------------------------------------------ frame geometry (what torchvision did for the reference)

def _fit_short_side(img, size):
    """torchvision's Resize: an int scales the SHORTER side to `size` keeping the aspect ratio; a pair is an explicit (height, width)"""
    if isinstance(size, (tuple, list)):
        return img.resize((int(size[1]), int(size[0])), Image.BILINEAR)
    width, height = img.size
    short, long_ = (width, height) if width <= height else (height, width)
    if short == size:
        return img
    scaled_long = int(size * long_ / short)
    target = (size, scaled_long) if width <= height else (scaled_long, size)
    return img.resize(target, Image.BILINEAR)


def _crop_middle(img, size):
    """torchvision's CenterCrop (zero padding first when the image is smaller than the crop)"""
    crop_h, crop_w = pair(size)
    width, height = img.size
    if width < crop_w or height < crop_h:
        padded = Image.new(img.mode, (max(width, crop_w), max(height, crop_h)))
        padded.paste(img, ((padded.size[0] - width) // 2, (padded.size[1] - height) // 2))
        img = padded
        width, height = img.size
    x0 = int(round((width - crop_w) / 2.0))
    y0 = int(round((height - crop_h) / 2.0))
    return img.crop((x0, y0, x0 + crop_w, y0 + crop_h))


def to_tensor(img):
    """PIL image -> (c, h, w) f32 in [0, 1]"""
    pixels = np.array(img, dtype=np.uint8)
    if pixels.ndim == 2:
        pixels = pixels[..., None]
    return torch.from_numpy(pixels).movedim(-1, 0).to(torch.float32) / 255.0


def to_pil_image(frame):
    """(c, h, w) tensor -> PIL image; floats are taken as [0, 1]"""
    if frame.is_floating_point():
        frame = (frame.detach().cpu() * 255).to(torch.uint8)
    hwc = frame.cpu().movedim(0, -1).numpy()
    channels = hwc.shape[-1]
    assert channels in _PIL_MODE, f'channels {channels} invalid'
    return Image.fromarray(hwc[..., 0] if channels == 1 else hwc, mode=_PIL_MODE[channels])


class FrameTransform:
    """resize -> optional random mirror -> centre crop -> tensor, per frame"""

    def __init__(self, image_size, mirror=False, force_rgb=False):
        self.image_size, self.mirror, self.force_rgb = image_size, mirror, force_rgb

    def __call__(self, img):
        if self.force_rgb and img.mode != 'RGB':
            img = img.convert('RGB')
        img = _fit_short_side(img, self.image_size)
        if self.mirror and random.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return to_tensor(_crop_middle(img, self.image_size))


def cast_num_frames(t, *, frames):
    """(c, f, h, w) with exactly `frames` frames: surplus frames dropped, missing ones zero-filled"""
    have = t.shape[1]
    if have >= frames:
        return t[:, :frames]
    filler = t.new_zeros((t.shape[0], frames - have, *t.shape[2:]))
    return torch.cat((t, filler), dim=1)


# ------------------------------------------------------------------------------------------ GIF

def seek_all_images(img, channels=3):
    """every frame of an (animated) PIL image, converted to the mode of `channels`"""
    assert channels in _PIL_MODE, f'channels {channels} invalid'
    for index in range(getattr(img, 'n_frames', 1)):
        img.seek(index)
        yield img.convert(_PIL_MODE[channels])


def video_tensor_to_gif(tensor, path, duration=120, loop=0, optimize=True):
    """(c, f, h, w) in [0, 1] -> animated GIF at `path`; returns the PIL frames"""
    pil_frames = [to_pil_image(tensor[:, f]) for f in range(tensor.shape[1])]
    pil_frames[0].save(path, save_all=True, append_images=pil_frames[1:], duration=duration, loop=loop, optimize=optimize)
    return pil_frames


def gif_to_tensor(path, channels=3, transform=to_tensor):
    """animated GIF -> (c, f, h, w)"""
    with Image.open(path) as img:
        return torch.stack([transform(frame) for frame in seek_all_images(img, channels=channels)], dim=1)


# ------------------------------------------------------------------------------------------ MP4 (OpenCV, when present)

def _opencv():
    try:
        import cv2
    except ImportError as err:
        raise ImportError('MP4 read / write goes through OpenCV (cv2), which is not installed here; GIFs need only PIL') from err
    return cv2


def crop_center(img, cropx, cropy):
    """centre window of an (h, w, c) array"""
    height, width = img.shape[:2]
    top, left = height // 2 - cropy // 2, width // 2 - cropx // 2
    return img[top: top + cropy, left: left + cropx]


def video_to_tensor(path, num_frames=-1, crop_size=None):
    """MP4 -> (channels, frames, height, width) f32 with OpenCV's 0..255 BGR values, the final decoded frame left out and the frame axis
    cut with `[:num_frames]`, all as the reference does (data.py:132-160)"""
    cv2 = _opencv()
    capture = cv2.VideoCapture(path)
    decoded = []
    while True:
        ok, frame = capture.read()
        if not ok:
            break
        decoded.append(crop_center(frame, *pair(crop_size)) if exists(crop_size) else frame)
    capture.release()
    clip = np.stack(decoded[:-1], axis=0)                               # (f, h, w, c)
    return torch.from_numpy(clip).movedim(-1, 0).float()[:, :num_frames]


def tensor_to_video(tensor, path, fps=25, video_format='MP4V'):
    """(c, f, h, w) with 0..255 values -> MP4 at `path`"""
    cv2 = _opencv()
    clip = tensor.detach().cpu()
    frames, height, width = clip.shape[-3:]
    writer = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*video_format), fps, (width, height))
    for f in range(frames):
        writer.write(clip[:, f].movedim(0, -1).numpy().astype(np.uint8))
    writer.release()
    return writer


# ------------------------------------------------------------------------------------------ folder datasets

class _FolderDataset(torch_data.Dataset):
    """every file below `folder` with one of the extensions `exts`"""

    def __init__(self, folder, exts):
        super().__init__()
        self.folder = folder
        root = Path(f'{folder}')
        self.paths = [p for ext in exts for p in root.glob(f'**/*.{ext}')]

    def __len__(self):
        return len(self.paths)


class ImageDataset(_FolderDataset):
    """images as (3, image_size, image_size) tensors: RGB, shorter side resized, random mirror, centre crop"""

    def __init__(self, folder, image_size, exts=['jpg', 'jpeg', 'png']):
        super().__init__(folder, exts)
        self.image_size = image_size
        print(f'{len(self.paths)} training samples found at {folder}')
        self.transform = FrameTransform(image_size, mirror=True, force_rgb=True)

    def __getitem__(self, index):
        with Image.open(self.paths[index]) as img:
            return self.transform(img)


class VideoDataset(_FolderDataset):
    """GIFs / MP4s as (channels, num_frames, image_size, image_size) tensors"""

    def __init__(self, folder, image_size, channels=3, num_frames=17, horizontal_flip=False, force_num_frames=True, exts=['gif', 'mp4']):
        super().__init__(folder, exts)
        self.image_size, self.channels = image_size, channels
        self.num_frames = num_frames if force_num_frames else None
        self.transform = FrameTransform(image_size, mirror=horizontal_flip)

    def __getitem__(self, index):
        path = self.paths[index]
        kind = path.suffix.lower()
        if kind == '.gif':
            clip = gif_to_tensor(path, channels=self.channels, transform=self.transform)
        elif kind == '.mp4':
            clip = video_to_tensor(str(path), crop_size=self.image_size)
        else:
            raise ValueError(f'unknown extension {path.suffix}')
        return clip if self.num_frames is None else cast_num_frames(clip, frames=self.num_frames)


# ------------------------------------------------------------------------------------------ batches of tensors and strings

def collate_tensors_and_strings(samples):
    """a list of tensors -> (stacked,); a list of tuples -> per position a stacked tensor or a list of strings"""
    if all(torch.is_tensor(s) for s in samples):
        return (torch.stack(samples, dim=0),)
    columns = []
    for column in zip(*samples):
        if all(torch.is_tensor(c) for c in column):
            columns.append(torch.stack(column, dim=0))
        elif all(isinstance(c, str) for c in column):
            columns.append(list(column))
        else:
            raise ValueError('detected invalid type being passed from dataset')
    return tuple(columns)


def DataLoader(*args, **kwargs):
    return torch_data.DataLoader(*args, collate_fn=collate_tensors_and_strings, **kwargs)

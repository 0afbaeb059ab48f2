"""Image / video I/O and datasets of the reference (`/root/reference/phenaki_pytorch/data.py:48-265`), host side, on PIL + numpy + torch
(this build depends on neither torchvision nor OpenCV): folders of images / GIFs as datasets of `(c, h, w)` / `(c, f, h, w)` f32 tensors in
[0, 1], GIF read / write for sampled videos, the string-aware collate of the trainers' DataLoader.  MP4 read / write goes through OpenCV
exactly as in the reference when `cv2` is importable and says so when it is not.  SURVEY.md 8f row 4 (data formats either side of the path).
"""
import random
from functools import partial
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import DataLoader as PytorchDataLoader
from torch.utils.data import Dataset


def exists(val):
    return val is not None


def identity(t, *args, **kwargs):
    return t


def pair(val):
    return val if isinstance(val, tuple) else (val, val)


def cast_num_frames(t, *, frames):
    """(c, f, h, w): cut or zero-pad the frame axis to `frames` (data.py:30-39)"""
    f = t.shape[1]
    if f == frames:
        return t
    if f > frames:
        return t[:, :frames]
    return F.pad(t, (0, 0, 0, 0, 0, frames - f))


# ---- the torchvision transforms the reference composes (Resize -> [RandomHorizontalFlip] -> CenterCrop -> ToTensor), on PIL

def _resize(img, size):
    """T.Resize: an int scales the SHORTER side to it (aspect kept, bilinear + antialias); a pair is (h, w)"""
    if isinstance(size, (tuple, list)):
        return img.resize((size[1], size[0]), Image.BILINEAR)
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        return img.resize((size, int(size * h / w)), Image.BILINEAR)
    return img.resize((int(size * w / h), size), Image.BILINEAR)


def _center_crop(img, size):
    ch, cw = pair(size)
    w, h = img.size
    if w < cw or h < ch:                                     # torchvision pads with zeros first
        canvas = Image.new(img.mode, (max(w, cw), max(h, ch)))
        canvas.paste(img, ((max(w, cw) - w) // 2, (max(h, ch) - h) // 2))
        img, (w, h) = canvas, canvas.size
    left, top = int(round((w - cw) / 2.)), int(round((h - ch) / 2.))
    return img.crop((left, top, left + cw, top + ch))


def to_tensor(img):
    """T.ToTensor: PIL image -> (c, h, w) f32 in [0, 1]"""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div_(255.)


def to_pil_image(t):
    """T.ToPILImage for a (c, h, w) float tensor in [0, 1] (or a uint8 one)"""
    if t.is_floating_point():
        t = t.detach().cpu().mul(255).byte()
    a = t.cpu().permute(1, 2, 0).numpy()
    return Image.fromarray(a[:, :, 0], mode='L') if a.shape[2] == 1 else Image.fromarray(a, mode=CHANNELS_TO_MODE[a.shape[2]])


class _Transform:
    def __init__(self, image_size, flip, to_rgb):
        self.image_size, self.flip, self.to_rgb = image_size, flip, to_rgb

    def __call__(self, img):
        if self.to_rgb and img.mode != 'RGB':
            img = img.convert('RGB')
        img = _resize(img, self.image_size)
        if self.flip and random.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        return to_tensor(_center_crop(img, self.image_size))


class ImageDataset(Dataset):
    """data.py:48-78: every jpg / jpeg / png under `folder` as a (3, image_size, image_size) tensor (random horizontal flip)"""

    def __init__(self, folder, image_size, exts=['jpg', 'jpeg', 'png']):
        super().__init__()
        self.folder = folder
        self.image_size = image_size
        self.paths = [p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}')]
        print(f'{len(self.paths)} training samples found at {folder}')
        self.transform = _Transform(image_size, flip=True, to_rgb=True)

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        return self.transform(Image.open(self.paths[index]))


# ---- GIF <-> (channels, frames, height, width) tensor (data.py:84-128)

CHANNELS_TO_MODE = {1: 'L', 3: 'RGB', 4: 'RGBA'}


def seek_all_images(img, channels=3):
    assert channels in CHANNELS_TO_MODE, f'channels {channels} invalid'
    mode = CHANNELS_TO_MODE[channels]
    i = 0
    while True:
        try:
            img.seek(i)
            yield img.convert(mode)
        except EOFError:
            break
        i += 1


def video_tensor_to_gif(tensor, path, duration=120, loop=0, optimize=True):
    images = list(map(to_pil_image, tensor.unbind(dim=1)))
    first_img, *rest_imgs = images
    first_img.save(path, save_all=True, append_images=rest_imgs, duration=duration, loop=loop, optimize=optimize)
    return images


def gif_to_tensor(path, channels=3, transform=to_tensor):
    img = Image.open(path)
    tensors = tuple(map(transform, seek_all_images(img, channels=channels)))
    return torch.stack(tensors, dim=1)


# ---- MP4 through OpenCV, as the reference does (data.py:132-195)

def _cv2():
    try:
        import cv2
        return cv2
    except ImportError as e:
        raise ImportError('MP4 read / write goes through OpenCV (cv2), which is not installed here; GIFs need only PIL') from e


def crop_center(img, cropx, cropy):
    y, x, c = img.shape
    startx = x // 2 - cropx // 2
    starty = y // 2 - cropy // 2
    return img[starty:(starty + cropy), startx:(startx + cropx), :]


def video_to_tensor(path, num_frames=-1, crop_size=None):
    """-> (channels, frames, height, width) f32 (0..255, as in the reference: data.py:158-160)"""
    cv2 = _cv2()
    video = cv2.VideoCapture(path)
    frames = []
    check = True
    while check:
        check, frame = video.read()
        if not check:
            continue
        if exists(crop_size):
            frame = crop_center(frame, *pair(crop_size))
        frames.append(frame[None])
    frames = np.array(np.concatenate(frames[:-1], axis=0))            # (the reference drops the last frame: data.py:155)
    frames_torch = torch.tensor(frames).permute(3, 0, 1, 2).float()
    return frames_torch[:, :num_frames, :, :]


def tensor_to_video(tensor, path, fps=25, video_format='MP4V'):
    cv2 = _cv2()
    tensor = tensor.cpu()
    num_frames, height, width = tensor.shape[-3:]
    fourcc = cv2.VideoWriter_fourcc(*video_format)
    video = cv2.VideoWriter(path, fourcc, fps, (width, height))
    for idx in range(num_frames):
        video.write(np.uint8(tensor[:, idx, :, :].permute(1, 2, 0).numpy()))
    video.release()
    return video


class VideoDataset(Dataset):
    """data.py:199-243: every gif / mp4 under `folder` as a (channels, num_frames, image_size, image_size) tensor"""

    def __init__(self, folder, image_size, channels=3, num_frames=17, horizontal_flip=False, force_num_frames=True, exts=['gif', 'mp4']):
        super().__init__()
        self.folder = folder
        self.image_size = image_size
        self.channels = channels
        self.paths = [p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}')]
        self.transform = _Transform(image_size, flip=horizontal_flip, to_rgb=False)
        self.gif_to_tensor = partial(gif_to_tensor, channels=self.channels, transform=self.transform)
        self.mp4_to_tensor = partial(video_to_tensor, crop_size=self.image_size)
        self.cast_num_frames_fn = partial(cast_num_frames, frames=num_frames) if force_num_frames else identity

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        path = self.paths[index]
        ext = path.suffix
        if ext == '.gif':
            tensor = self.gif_to_tensor(path)
        elif ext == '.mp4':
            tensor = self.mp4_to_tensor(str(path))
        else:
            raise ValueError(f'unknown extension {ext}')
        return self.cast_num_frames_fn(tensor)


# ---- DataLoader that can collate strings beside tensors (data.py:247-268)

def collate_tensors_and_strings(data):
    if all(isinstance(d, torch.Tensor) for d in data):
        return (torch.stack(data, dim=0),)
    output = []
    for datum in zip(*data):
        if all(isinstance(d, torch.Tensor) for d in datum):
            datum = torch.stack(datum, dim=0)
        elif all(isinstance(d, str) for d in datum):
            datum = list(datum)
        else:
            raise ValueError('detected invalid type being passed from dataset')
        output.append(datum)
    return tuple(output)


def DataLoader(*args, **kwargs):
    return PytorchDataLoader(*args, collate_fn=collate_tensors_and_strings, **kwargs)

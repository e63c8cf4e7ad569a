"""speech2lip_amd -- MI355X-native lip-render hot path of Speech2Lip."""
from .config import get_model, get_trainer, load_config, may_config, method_dict
from .data import ClipStreamer, ClipTensors, FramePrefetcher, FrameWriter, SomeonesLipClip, from8b, render_clip_frames, to8b, write_frames
from .rendering import get_coords
from .talking_face import Embedder, FrameGraph, PositionalEncodingTime, TalkingFace
from . import training
from .training import LipTrainStep, StageOneStep, SyncChain, Trainer, predict_lip_image
from . import autograd
from .unet import SimpleUnetLight
from .syncnet import SyncLoss, SyncNet_color
from .lpips import LPIPS
from . import geometry
from . import optim
from .optim import FusedAdam

__all__ = ["TalkingFace", "FrameGraph", "Embedder", "PositionalEncodingTime", "get_coords", "load_config", "may_config", "get_model", "get_trainer", "method_dict", "Trainer",
           "predict_lip_image", "LipTrainStep", "StageOneStep", "SyncChain", "training", "autograd",
           "SimpleUnetLight", "SomeonesLipClip", "ClipTensors", "render_clip_frames", "write_frames", "to8b", "from8b", "ClipStreamer", "FrameWriter", "FramePrefetcher", "SyncNet_color", "SyncLoss", "LPIPS", "geometry", "optim", "FusedAdam"]

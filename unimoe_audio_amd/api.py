"""High-level API surface of the reference, on the HIP engine.

Keeps both spellings the reference ships: the HF-directory flavour (`UniMoE_Audio.py:39-261`:
`text_to_speech(transcription, prompt_transcription, prompt_wav, output_dir, max_audio_seconds, ...)`) and the in-repo twin
(`utils/UniMoE_Audio_mod.py:294-619`: `caption`, `prompt_text`, `save_name`, `cfg_scale`, ...).  Prompt templates, negative /
positive prompt pairing and `max_tokens = 50 * seconds` follow the reference (mod.py:56-59,343-348,449-466;
UniMoE_Audio.py:137-138).

The tokenizer (HF files under `model_path`) and the DAC weights (weights_16khz.pth) are assets that are not available offline; they
are loaded lazily and a clear error is raised when they are missing.  The codec itself runs on the HIP path (unimoe_audio_amd/dac.py).
tests/test_gpu_api.py drives text_to_speech / text_to_music end to end with a stand-in tokenizer and a randomly initialised codec.
"""
from __future__ import annotations

import os
from typing import List, Optional, Union

import torch

from .codec_utils import DecoderOutput, generate_output, prepare_audio_prompt, preprocess_codec
from .config import UniMoEAudioConfig
from .model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration

SYSTEM_MESSAGE = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
INPUT_FORMAT = "<|im_start|>user\n{}<|im_end|>\n<|im_start|>assistant\n"
AUDIO_START = "<|AUDIO_START|>"


class UniMoEAudio:
    def __init__(self, model_path: Optional[str], device_id: int = 0, config: Optional[UniMoEAudioConfig] = None,
                 model: Optional[UniAudioRVQQwen2_5VLMoEForConditionalGeneration] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("UniMoEAudio needs a ROCm device: the accelerated path has no CPU fallback")
        torch.cuda.set_device(device_id)
        self.device = torch.device(f"cuda:{device_id}")
        self.TORCH_DTYPE = torch.bfloat16
        self.model_path = model_path
        if model is not None:
            self.model = model
        else:
            cfg = config
            if cfg is None:
                if not model_path or not os.path.exists(os.path.join(model_path, "config.json")):
                    raise FileNotFoundError("model_path must contain the reference config.json (and the safetensors shards)")
                cfg = UniMoEAudioConfig.from_json(os.path.join(model_path, "config.json"))
            self.model = UniAudioRVQQwen2_5VLMoEForConditionalGeneration(cfg)
            self._load_weights(model_path)
            self.model = self.model.to(self.device, torch.bfloat16).eval()
        self._tokenizer = None
        self._dac = None

    # ---- third-party assets ---------------------------------------------------------------------------------------
    def _load_weights(self, model_path):
        from . import checkpoint
        checkpoint.load_checkpoint(self.model, model_path or "")      # HF shards (+ index), reference key spelling, streamed

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from transformers import AutoTokenizer
            self._tokenizer = AutoTokenizer.from_pretrained(self.model_path, padding_side="left", use_fast=False)
        return self._tokenizer

    @property
    def dac(self):
        """The DAC codec on the HIP path (unimoe_audio_amd/dac.py; reference utils/UniMoE_Audio_utils.py:56-134).  Weights are looked
        for like the reference does (DAC_WEIGHTS, <model_path>/dac_model/weights_16khz.pth, ...); FileNotFoundError when absent."""
        if self._dac is None:
            from .dac import Dac
            cand = None
            if self.model_path:
                for p in (os.path.join(self.model_path, "dac_model", "weights_16khz.pth"), os.path.join(self.model_path, "weights_16khz.pth")):
                    if os.path.isfile(p):
                        cand = p
                        break
            self._dac = Dac(cand, device=self.device)
        return self._dac

    @dac.setter
    def dac(self, codec):
        self._dac = codec

    # ---- the accelerated part: tokens in, codes out -----------------------------------------------------------------
    @torch.no_grad()
    def generate_codes(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, codec_input_ids: Optional[torch.Tensor] = None,
                       max_audio_seconds: int = 10, min_audio_seconds: int = 2, cfg_scale: float = 3.0, temperature: float = 1.2,
                       top_p: float = 0.95, cfg_filter_top_k: int = 45, eos_prob_mul_factor: float = 0.8, do_sample: bool = True,
                       seed: int = 0):
        """input_ids [2B, T] (negative prompt, positive prompt per sample), returns the list of [len_i, 12] code tensors
        exactly as the reference's `generate_output` does (delay pattern reverted)."""
        cfg = self.model.config
        B = input_ids.shape[0] // 2
        prefill, steps = prepare_audio_prompt(cfg, [None] * B)
        dec = DecoderOutput(prefill, steps, self.device)
        codes, lengths = self.model.generate(input_ids, attention_mask, dec, max_tokens=max_audio_seconds * 50,
                                             min_tokens=min_audio_seconds * 50, codec_input_ids=codec_input_ids,
                                             cfg_scale=cfg_scale, temperature=temperature, top_p=top_p,
                                             cfg_filter_top_k=cfg_filter_top_k, eos_prob_mul_factor=eos_prob_mul_factor,
                                             do_sample=do_sample, seed=seed)
        if codes is None:
            return []
        return generate_output(cfg, codes, lengths)

    # ---- reference task methods (both spellings) -----------------------------------------------------------------------
    def _texts(self, obj: Union[str, List[str]]) -> List[str]:
        if isinstance(obj, str):
            obj = [obj]
        obj = [c for c in obj if c.strip()]
        if not obj:
            raise ValueError("Please enter valid target texts.")        # UniMoE_Audio.py:94-103
        return obj

    def _finish(self, audios, output_dir, stem):
        os.makedirs(output_dir, exist_ok=True)
        paths = []
        for i, a in enumerate(audios):
            path = os.path.join(output_dir, f"generated_{stem}_{i}.wav")
            self.dac.decode(a.transpose(0, 1).unsqueeze(0), save_path=path, min_duration=1)
            paths.append(path)
        return paths

    def text_to_music(self, caption: Union[str, List[str]], output_dir: str = "./", max_audio_seconds: int = 20,
                      min_audio_seconds: int = 8, temperature: float = 1.0, top_p: float = 1.0, cfg_filter_top_k: int = 45,
                      save_name: str = "music", cfg_scale: float = 10.0, eos_prob_mul_factor: float = 0.6, do_sample: bool = True,
                      **_) -> List[str]:
        caption = self._texts(caption)
        neg = SYSTEM_MESSAGE + INPUT_FORMAT.format("<|MUSIC_START|>Low quality.<|MUSIC_END|>") + AUDIO_START
        texts = []
        for c in caption:
            texts += [neg, SYSTEM_MESSAGE + INPUT_FORMAT.format("<|MUSIC_START|>" + c + "<|MUSIC_END|>") + AUDIO_START]
        enc = self.tokenizer(texts, add_special_tokens=False, return_tensors="pt", padding=True)
        audios = self.generate_codes(enc.input_ids, enc.attention_mask, None, max_audio_seconds, min_audio_seconds, cfg_scale,
                                     temperature, top_p, cfg_filter_top_k, eos_prob_mul_factor, do_sample)
        return self._finish(audios, output_dir, save_name)

    def text_to_speech(self, transcription: Union[str, List[str], None] = None, prompt_transcription: Optional[str] = None,
                       prompt_wav: Optional[str] = None, output_dir: str = "./", max_audio_seconds: int = 10,
                       min_audio_seconds: int = 2, temperature: float = 1.0, top_p: float = 1.0, cfg_filter_top_k: int = 45,
                       caption=None, prompt_text=None, prompt_codec=None, save_name: str = "speech", cfg_scale: float = 1.0,
                       eos_prob_mul_factor: float = 1.0, do_sample: bool = True, **_) -> List[str]:
        texts_in = self._texts(transcription if transcription is not None else caption)
        ptxt = prompt_transcription if prompt_transcription is not None else prompt_text
        if prompt_codec is None:
            if prompt_wav is None:
                raise ValueError("Please provide a reference audio file.")
            prompt_codec = self.dac.encode(prompt_wav)
        cfg = self.model.config
        pc = preprocess_codec(cfg, prompt_codec)
        prompt_caption = ("<|SPEECH_PROMPT_START|>" + ptxt + "<|SPEECH_PROMPT_END|>" + "<|VOICE_PROMPT_START|>" +
                          "<|AUDIO_PLACEHOLDER|>" * pc.shape[0] + "<|VOICE_PROMPT_END|>")
        wrap = lambda x: prompt_caption + "<|SPEECH_START|>" + x + "<|SPEECH_END|>"   # noqa: E731
        texts = []
        for t in texts_in:
            texts += [SYSTEM_MESSAGE + INPUT_FORMAT.format(wrap("")) + AUDIO_START,
                      SYSTEM_MESSAGE + INPUT_FORMAT.format(wrap(t)) + AUDIO_START]
        enc = self.tokenizer(texts, add_special_tokens=False, return_tensors="pt", padding=True)
        codec = pc.unsqueeze(0).expand(len(texts), -1, -1).reshape(-1, pc.shape[1])
        audios = self.generate_codes(enc.input_ids, enc.attention_mask, codec, max_audio_seconds, min_audio_seconds, cfg_scale,
                                     temperature, top_p, cfg_filter_top_k, eos_prob_mul_factor, do_sample)
        return self._finish(audios, output_dir, save_name)

    def video_text_to_music(self, video, caption: Union[str, List[str]], output_dir: str = "./", max_audio_seconds: int = 20,
                            min_audio_seconds: int = 8, temperature: float = 1.0, top_p: float = 1.0, cfg_filter_top_k: int = 45,
                            save_name: str = "video_music", cfg_scale: float = 10.0, eos_prob_mul_factor: float = 0.6, do_sample: bool = True,
                            fps: float = 1.0, sampling_fps: float = 1.0, max_frames: int = 8, vision_in_generate: bool = False, **_) -> List[str]:
        """reference UniMoE_Audio.py:203-257 / utils/UniMoE_Audio_mod.py:483-619: a video (here: its frames, a uint8 / float tensor
        [F, H, W, 3] or [F, 3, H, W]; file decoding needs moviepy / qwen_vl_utils, absent offline) + a caption -> music.  The frames are
        resized to multiples of 28 px within the reference's pixel budget (mod.py:49-53: at most 64 * 28 * 28 per frame), cut into the
        processor's patch layout; their tokens sit between <|vision_start|> and <|vision_end|>.
        vision_in_generate: False (default) = the reference's inference path, whose generate() never feeds the pixels to the model (the
        pad tokens keep their text embeddings, positions stay 1-D; see model.generate); True = vision tower + 3-D positions.
        A file path is decoded when a decoder is importable here (torchvision.io / decord / moviepy, in that order)."""
        from .vision import frames_to_patches
        caption = self._texts(caption)
        if isinstance(video, (str, bytes, os.PathLike)):
            video = _decode_video_file(os.fspath(video), sampling_fps, max_frames)
        frames = video if torch.is_tensor(video) else torch.as_tensor(video)
        if frames.dim() != 4:
            raise ValueError("video: pass the frames as a [F, H, W, 3] or [F, 3, H, W] tensor (file decoding is not available offline)")
        patches, grid = frames_to_patches(frames[:max_frames], max_pixels=64 * 28 * 28)
        n_tok = int(grid.prod()) // 4
        vid = "<|vision_start|>" + "<|video_pad|>" * n_tok + "<|vision_end|>"
        neg = SYSTEM_MESSAGE + INPUT_FORMAT.format(vid + "<|MUSIC_START|>Low quality.<|MUSIC_END|>") + AUDIO_START
        texts = []
        for c in caption:
            texts += [neg, SYSTEM_MESSAGE + INPUT_FORMAT.format(vid + "<|MUSIC_START|>" + c + "<|MUSIC_END|>") + AUDIO_START]
        enc = self.tokenizer(texts, add_special_tokens=False, return_tensors="pt", padding=True)
        R = len(texts)
        cfg = self.model.config
        prefill, steps = prepare_audio_prompt(cfg, [None] * (R // 2))
        dec = DecoderOutput(prefill, steps, self.device)
        codes, lengths = self.model.generate(enc.input_ids, enc.attention_mask, dec, max_tokens=max_audio_seconds * 50,
                                             min_tokens=min_audio_seconds * 50, pixel_values_videos=patches.repeat(R, 1),
                                             video_grid_thw=grid[None].repeat(R, 1), second_per_grid_ts=torch.full((R,), 2.0 / max(sampling_fps, 1e-6)),
                                             cfg_scale=cfg_scale, temperature=temperature, top_p=top_p, cfg_filter_top_k=cfg_filter_top_k,
                                             eos_prob_mul_factor=eos_prob_mul_factor, do_sample=do_sample, vision_in_generate=vision_in_generate)
        audios = [] if codes is None else generate_output(cfg, codes, lengths)
        return self._finish(audios, output_dir, save_name)


def _decode_video_file(path: str, sampling_fps: float, max_frames: int) -> torch.Tensor:
    """Frames [F, H, W, 3] uint8 of a video file, sampled at `sampling_fps` (reference utils/UniMoE_Audio_mod.py:158-213 uses moviepy +
    qwen_vl_utils).  Tries the decoders that may be importable here; raises a ValueError that says what to pass instead when none is."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    errors = []
    try:
        from torchvision.io import read_video          # type: ignore
        frames, _, info = read_video(path, pts_unit="sec", output_format="THWC")
        step = max(int(round(float(info.get("video_fps", 1.0)) / max(sampling_fps, 1e-6))), 1)
        return frames[::step][:max_frames]
    except Exception as e:      # not installed, or no backend for the container
        errors.append(f"torchvision.io: {e!r}")
    try:
        import decord                                   # type: ignore
        vr = decord.VideoReader(path)
        step = max(int(round(vr.get_avg_fps() / max(sampling_fps, 1e-6))), 1)
        idx = list(range(0, len(vr), step))[:max_frames]
        return torch.from_numpy(vr.get_batch(idx).asnumpy())
    except Exception as e:
        errors.append(f"decord: {e!r}")
    try:
        from moviepy.editor import VideoFileClip       # type: ignore
        import numpy as np
        clip = VideoFileClip(path)
        ts = [i / max(sampling_fps, 1e-6) for i in range(max_frames) if i / max(sampling_fps, 1e-6) < clip.duration]
        return torch.from_numpy(np.stack([clip.get_frame(t) for t in ts]))
    except Exception as e:
        errors.append(f"moviepy: {e!r}")
    raise ValueError("video: no video decoder is importable here (" + "; ".join(errors) + "): pass the frames as a [F, H, W, 3] or "
                     "[F, 3, H, W] tensor instead")


def create_unimoe_audio(model_path: str, device_id: int = 0) -> UniMoEAudio:
    return UniMoEAudio(model_path, device_id)

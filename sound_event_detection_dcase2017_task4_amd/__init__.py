"""MI355X-native (gfx950) hot path of qiuqiangkong/sound_event_detection_dcase2017_task4.

Layout mirrors the reference so it drops in:
    sound_event_detection_dcase2017_task4_amd/pytorch/{models,losses,pytorch_utils,main}.py
    sound_event_detection_dcase2017_task4_amd/utils/{config,utilities,data_generator}.py
Compute lives in csrc/*.hip behind the C ABI of include/sed_hip.h (libsed_hip.so).
"""
__version__ = "0.1.0"

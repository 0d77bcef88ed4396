"""Task constants under the reference's names (utils/config.py): audio format, log-mel front-end parameters and the 17
DCASE 2017 Task 4 classes.  The class table is kept as one (AudioSet id, label, training clips) row per class; the
reference's parallel lists are derived from it."""

# audio: 10 s mono clips at 32 kHz
sample_rate = 32000
audio_duration = 10
audio_samples = audio_duration * sample_rate

# front-end (PANNs settings): 1024-point Hann STFT, hop 320 -> 100 frames/s, 64 mel bands over 50 Hz .. 14 kHz
window_size, hop_size = 1024, 320
mel_bins, fmin, fmax = 64, 50, 14000
frames_per_second = sample_rate // hop_size
window, center, pad_mode = 'hann', True, 'reflect'
ref, amin, top_db = 1.0, 1e-10, None
device = 'cuda'

_CLASS_TABLE = (
    ('/m/0284vy3', 'Train horn', 441),
    ('/m/05x_td', 'Air horn, truck horn', 407),
    ('/m/02mfyn', 'Car alarm', 273),
    ('/m/02rhddq', 'Reversing beeps', 337),
    ('/m/0199g', 'Bicycle', 624),
    ('/m/06_fw', 'Skateboard', 2399),
    ('/m/012n7d', 'Ambulance (siren)', 2399),
    ('/m/012ndj', 'Fire engine, fire truck (siren)', 1506),
    ('/m/0dgbq', 'Civil defense siren', 744),
    ('/m/04qvtq', 'Police car (siren)', 2020),
    ('/m/03qc9zr', 'Screaming', 1617),
    ('/m/0k4j', 'Car', 25744),
    ('/t/dd00134', 'Car passing by', 3724),
    ('/m/01bjv', 'Bus', 3745),
    ('/m/07r04', 'Truck', 7090),
    ('/m/04_sv', 'Motorcycle', 3291),
    ('/m/07jdr', 'Train', 2301),
)
ids = [row[0] for row in _CLASS_TABLE]
labels = [row[1] for row in _CLASS_TABLE]
samples_num = [row[2] for row in _CLASS_TABLE]
classes_num = len(_CLASS_TABLE)
lb_to_idx = dict((name, k) for k, name in enumerate(labels))
idx_to_lb = dict(enumerate(labels))

"""Host-side mirror of the reference facade (cv::cuda::EfficientFeatures, cv::cuda::BAD, cv::cuda::HashSIFT)
over the C ABI of libefx_hip.so (include/efx.h).

Reference interface: modules/cuda_efficient_features/include/cuda_efficient_features.h:28-98 and
cuda_efficient_descriptors.h:27-126.  Method names, argument meaning and error behaviour follow it
(bad arguments raise, like CV_Assert / CV_Error); device buffers are torch CUDA tensors, which play the
role of cv::cuda::GpuMat, and host images / keypoints are numpy arrays (cv::Mat / std::vector<KeyPoint>).

There is NO CPU fallback: importing works without a GPU (so the ABI can be inspected), but creating a
detector or describer raises EfxError when libefx_hip.so or a HIP device is missing.

The directory name contains '-', so import it through cef_loader.load() (repo root), which registers it
as module `cef_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libefx_hip.so")

EFX_OK = 0
STATUS_NAMES = {0: "EFX_OK", -1: "EFX_ERR_BAD_ARG", -2: "EFX_ERR_UNSUPPORTED", -3: "EFX_ERR_HIP",
                -4: "EFX_ERR_NO_DEVICE", -5: "EFX_ERR_NOMEM", -6: "EFX_ERR_OVERFLOW"}

# every symbol include/efx.h declares (checked by tests/test_abi.py)
EFX_ERR_OVERFLOW = -6

ABI_SYMBOLS = [
    "efx_default_params", "efx_create", "efx_destroy", "efx_device_bytes", "efx_trim_memory", "efx_cached_bytes", "efx_last_error", "efx_version",
    "efx_set_max_features", "efx_get_max_features", "efx_set_scale_factor", "efx_get_scale_factor",
    "efx_set_nlevels", "efx_get_nlevels", "efx_set_first_level", "efx_get_first_level",
    "efx_set_fast_threshold", "efx_get_fast_threshold", "efx_set_nonmax_radius", "efx_get_nonmax_radius",
    "efx_set_descriptor_type", "efx_get_descriptor_type",
    "efx_descriptor_size", "efx_descriptor_dtype", "efx_default_norm",
    "efx_detect_async", "efx_detect_and_compute_async", "efx_compute_async", "efx_compute_kp4_async",
    "efx_last_count", "efx_last_level_stats", "efx_overflow_events", "efx_tracked_streams",
    "efx_detect", "efx_compute", "efx_detect_and_compute", "efx_convert",
    "efx_bad_create", "efx_hashsift_create", "efx_describer_destroy", "efx_describer_descriptor_size",
    "efx_describer_last_error", "efx_describer_compute_kp4_async", "efx_describer_compute_async",
    "efx_describer_compute", "efx_describer_hashsift_debug_async",
    "efx_matcher_create", "efx_matcher_destroy", "efx_matcher_last_error", "efx_match_knn2_async",
    "efx_match_crosscheck_async",
    "efx_detect_and_compute_batch_async", "efx_detect_and_compute_masked_async", "efx_compute_provided_async", "efx_detect_and_compute_ex",
    "efx_ic_angles_async", "efx_ic_angles", "efx_descriptors_to_csv",
    "efx_cvt_gray_async", "efx_host_alloc", "efx_host_free", "efx_uploader_create", "efx_uploader_destroy",
    "efx_uploader_last_error", "efx_upload_gray_async", "efx_uploader_release", "efx_uploader_wait_uploaded",
    "efx_describer_compute_color",
    "efx_profile_enable", "efx_profile_set_stride", "efx_profile_set_groups", "efx_profile_read",
    "efx_level_geometry", "efx_copy_level_async",
]


class EfxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int), ("first_level", C.c_int),
                ("fast_threshold", C.c_int), ("nonmax_radius", C.c_int), ("descriptor_type", C.c_int)]


class LevelStats(C.Structure):
    _fields_ = [("n_candidates", C.c_int), ("n_after_nms", C.c_int), ("n_kept", C.c_int)]


# efx_keypoint / cv::KeyPoint as a numpy record
KEYPOINT_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                           ("response", np.float32), ("octave", np.int32), ("class_id", np.int32)])

_lib = None


def lib():
    """Loads libefx_hip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EfxError(-4, f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(make -C cuda-efficient-features_amd/csrc); there is no CPU fallback")
        try:
            import torch  # noqa: F401  -- must come first: the library then binds to the HIP runtime torch loaded,
        except ImportError:   # so that torch tensors and libefx_hip.so share one runtime instance (devices, streams)
            pass
        L = C.CDLL(LIB_PATH)
        L.efx_last_error.restype = C.c_char_p
        L.efx_last_error.argtypes = [C.c_void_p]
        L.efx_describer_last_error.restype = C.c_char_p
        L.efx_describer_last_error.argtypes = [C.c_void_p]
        L.efx_get_scale_factor.restype = C.c_float
        L.efx_get_scale_factor.argtypes = [C.c_void_p]
        L.efx_set_scale_factor.argtypes = [C.c_void_p, C.c_float]
        for name in ("efx_set_max_features", "efx_set_nlevels", "efx_set_first_level", "efx_set_fast_threshold",
                     "efx_set_nonmax_radius", "efx_set_descriptor_type"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        for name in ("efx_get_max_features", "efx_get_nlevels", "efx_get_first_level", "efx_get_fast_threshold",
                     "efx_get_nonmax_radius", "efx_get_descriptor_type", "efx_descriptor_size", "efx_descriptor_dtype",
                     "efx_default_norm", "efx_destroy"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.efx_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.efx_detect_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_int, C.c_void_p, C.c_void_p]
        L.efx_detect_and_compute_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                                   C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.efx_compute_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.efx_compute_kp4_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                            C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
        L.efx_last_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.efx_overflow_events.restype = C.c_int; L.efx_overflow_events.argtypes = [C.c_void_p]
        L.efx_tracked_streams.restype = C.c_int; L.efx_tracked_streams.argtypes = [C.c_void_p]
        L.efx_device_bytes.restype = C.c_size_t
        L.efx_trim_memory.restype = C.c_size_t; L.efx_trim_memory.argtypes = []
        L.efx_cached_bytes.restype = C.c_size_t; L.efx_cached_bytes.argtypes = []
        L.efx_device_bytes.argtypes = [C.c_void_p]
        L.efx_last_level_stats.argtypes = [C.c_void_p, C.POINTER(LevelStats), C.c_int, C.POINTER(C.c_int)]
        L.efx_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.efx_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.efx_detect_and_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_int, C.POINTER(C.c_int)]
        L.efx_convert.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.efx_bad_create.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_void_p)]
        L.efx_hashsift_create.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_void_p)]
        L.efx_describer_destroy.argtypes = [C.c_void_p]
        L.efx_describer_descriptor_size.argtypes = [C.c_void_p]
        L.efx_describer_compute_kp4_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                                      C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
        L.efx_describer_compute_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                                  C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.efx_describer_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_size_t]
        L.efx_describer_hashsift_debug_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                                         C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.efx_matcher_create.argtypes = [C.POINTER(C.c_void_p)]
        L.efx_matcher_destroy.argtypes = [C.c_void_p]
        L.efx_matcher_last_error.restype = C.c_char_p
        L.efx_matcher_last_error.argtypes = [C.c_void_p]
        for name in ("efx_match_knn2_async", "efx_match_crosscheck_async"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.efx_detect_and_compute_batch_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                                         C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.efx_detect_and_compute_masked_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.efx_compute_provided_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                                 C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.efx_detect_and_compute_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.efx_ic_angles_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.efx_ic_angles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.efx_descriptors_to_csv.restype = C.c_long
        L.efx_descriptors_to_csv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.c_size_t]
        L.efx_cvt_gray_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.efx_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.efx_host_free.argtypes = [C.c_void_p]
        L.efx_uploader_create.argtypes = [C.POINTER(C.c_void_p)]
        L.efx_uploader_destroy.argtypes = [C.c_void_p]
        L.efx_uploader_last_error.restype = C.c_char_p
        L.efx_uploader_last_error.argtypes = [C.c_void_p]
        L.efx_upload_gray_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        L.efx_uploader_release.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.efx_uploader_wait_uploaded.argtypes = [C.c_void_p, C.c_void_p]
        L.efx_describer_compute_color.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p,
                                                  C.c_int, C.c_void_p, C.c_size_t]
        L.efx_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.efx_profile_set_stride.argtypes = [C.c_void_p, C.c_int]
        L.efx_profile_set_groups.argtypes = [C.c_void_p, C.c_uint]
        L.efx_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        L.efx_level_geometry.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_float)]
        L.efx_copy_level_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def _stream_ptr(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _host_image(image):
    img = np.ascontiguousarray(image)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise EfxError(-1, "Image should be 8UC1")        # CV_Assert(_image.type() == CV_8U), .cpp:228
    return img


def _dev_image(image):
    import torch
    if not (isinstance(image, torch.Tensor) and image.is_cuda and image.dtype == torch.uint8 and image.dim() == 2
            and image.stride(1) == 1):
        raise EfxError(-1, "device image must be a 2-D uint8 CUDA tensor with unit column stride")
    return image


def keypoints_array(n):
    return np.zeros(n, dtype=KEYPOINT_DTYPE)


class EfficientFeatures:
    """cv::cuda::EfficientFeatures (cuda_efficient_features.h:28-98)."""
    LOCATION_ROW, RESPONSE_ROW, ANGLE_ROW, OCTAVE_ROW, SIZE_ROW, ROWS_COUNT = 0, 1, 2, 3, 4, 5
    BAD_256, BAD_512, HASH_SIFT_256, HASH_SIFT_512 = 0, 1, 2, 3

    def __init__(self, nfeatures=5000, scaleFactor=1.2, nlevels=8, firstLevel=0, fastThreshold=20, nonmaxRadius=15,
                 dtype=2):
        self._h = C.c_void_p()
        p = Params(nfeatures, scaleFactor, nlevels, firstLevel, fastThreshold, nonmaxRadius, dtype)
        rc = lib().efx_create(C.byref(p), C.byref(self._h))
        if rc != EFX_OK:
            self._h = C.c_void_p()
            raise EfxError(rc, lib().efx_last_error(None).decode())

    @staticmethod
    def create(nfeatures=5000, scaleFactor=1.2, nlevels=8, firstLevel=0, fastThreshold=20, nonmaxRadius=15, dtype=2):
        return EfficientFeatures(nfeatures, scaleFactor, nlevels, firstLevel, fastThreshold, nonmaxRadius, dtype)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().efx_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc != EFX_OK:
            raise EfxError(rc, lib().efx_last_error(self._h).decode())

    # ---- getters / setters (cuda_efficient_features.h:78-97) ----
    def setMaxFeatures(self, v): self._check(lib().efx_set_max_features(self._h, int(v)))
    def getMaxFeatures(self): return lib().efx_get_max_features(self._h)
    def setScaleFactor(self, v): self._check(lib().efx_set_scale_factor(self._h, float(v)))
    def getScaleFactor(self): return lib().efx_get_scale_factor(self._h)
    def setNLevels(self, v): self._check(lib().efx_set_nlevels(self._h, int(v)))
    def getNLevels(self): return lib().efx_get_nlevels(self._h)
    def setFirstLevel(self, v): self._check(lib().efx_set_first_level(self._h, int(v)))
    def getFirstLevel(self): return lib().efx_get_first_level(self._h)
    def setFastThreshold(self, v): self._check(lib().efx_set_fast_threshold(self._h, int(v)))
    def getFastThreshold(self): return lib().efx_get_fast_threshold(self._h)
    def setNonmaxRadius(self, v): self._check(lib().efx_set_nonmax_radius(self._h, int(v)))
    def getNonmaxRadius(self): return lib().efx_get_nonmax_radius(self._h)
    def setDescriptorType(self, v): self._check(lib().efx_set_descriptor_type(self._h, int(v)))
    def getDescriptorType(self): return lib().efx_get_descriptor_type(self._h)
    def descriptorSize(self): return lib().efx_descriptor_size(self._h)
    def descriptorType(self): return lib().efx_descriptor_dtype(self._h)
    def defaultNorm(self): return lib().efx_default_norm(self._h)

    # ---- asynchronous device API ----
    def detectAsync(self, image, keypoints=None, count=None, capacity=None, stream=None):
        """image: 2-D uint8 CUDA tensor.  Returns (keypoints 5 x capacity float32 tensor, count int32 tensor)."""
        import torch
        image = _dev_image(image)
        capacity = self.getMaxFeatures() if capacity is None else int(capacity)
        if keypoints is None:
            keypoints = torch.zeros((5, max(capacity, 1)), dtype=torch.float32, device=image.device)
        if count is None:
            count = torch.zeros(1, dtype=torch.int32, device=image.device)
        self._check(lib().efx_detect_async(self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0),
                                           keypoints.data_ptr(), keypoints.stride(0) * 4, capacity, count.data_ptr(),
                                           _stream_ptr(stream)))
        return keypoints, count

    def detectAndComputeAsync(self, image, keypoints=None, descriptors=None, count=None, capacity=None, stream=None,
                              mask=None, useProvidedKeypoints=False, n=None, want_descriptors=True):
        """detectAndComputeAsync(image, mask, keypoints, descriptors, useProvidedKeypoints, stream)
        (cuda_efficient_features.h:66-73).  mask: H x W uint8 CUDA tensor (spec S12); useProvidedKeypoints: `keypoints`
        (5 x N, first `n` columns) are described as detectAndCompute would have (spec S13) and only descriptors return."""
        import torch
        image = _dev_image(image)
        if useProvidedKeypoints:
            if keypoints is None or keypoints.dim() != 2 or keypoints.shape[0] != 5 or keypoints.dtype != torch.float32:
                raise EfxError(-1, "keypoints must be a 5 x N float32 matrix")
            n = keypoints.shape[1] if n is None else int(n)
            if descriptors is None:
                descriptors = torch.zeros((max(n, 1), self.descriptorSize()), dtype=torch.uint8, device=image.device)
            self._check(lib().efx_compute_provided_async(
                self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), keypoints.data_ptr(),
                keypoints.stride(0) * 4, n, descriptors.data_ptr(), descriptors.stride(0), _stream_ptr(stream)))
            return descriptors[:n]
        capacity = self.getMaxFeatures() if capacity is None else int(capacity)
        if keypoints is None:
            keypoints = torch.zeros((5, max(capacity, 1)), dtype=torch.float32, device=image.device)
        if descriptors is None and want_descriptors:
            descriptors = torch.zeros((max(capacity, 1), self.descriptorSize()), dtype=torch.uint8, device=image.device)
        if count is None:
            count = torch.zeros(1, dtype=torch.int32, device=image.device)
        if mask is not None:
            mask = _dev_image(mask)
            if tuple(mask.shape) != tuple(image.shape):
                raise EfxError(-1, "mask must have the size of the image")
            self._check(lib().efx_detect_and_compute_masked_async(
                self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), mask.data_ptr(), mask.stride(0),
                keypoints.data_ptr(), keypoints.stride(0) * 4, descriptors.data_ptr() if descriptors is not None else None,
                descriptors.stride(0) if descriptors is not None else 0, capacity, count.data_ptr(), _stream_ptr(stream)))
            return keypoints, descriptors, count
        if descriptors is None:          # want_descriptors=False: the detect-only entry point
            self._check(lib().efx_detect_async(
                self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), keypoints.data_ptr(),
                keypoints.stride(0) * 4, capacity, count.data_ptr(), _stream_ptr(stream)))
            return keypoints, None, count
        self._check(lib().efx_detect_and_compute_async(
            self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), keypoints.data_ptr(),
            keypoints.stride(0) * 4, descriptors.data_ptr(), descriptors.stride(0), capacity, count.data_ptr(),
            _stream_ptr(stream)))
        return keypoints, descriptors, count

    def computeAsync(self, image, keypoints, n=None, descriptors=None, stream=None):
        """keypoints: 5 x N float32 CUDA tensor (the detector's layout; size forced to 31)."""
        import torch
        image = _dev_image(image)
        if keypoints.dim() != 2 or keypoints.shape[0] != 5 or keypoints.dtype != torch.float32:
            raise EfxError(-1, "keypoints must be a 5 x N float32 matrix")    # CV_Assert(tmp.rows == 5 ...), .cpp:111
        n = keypoints.shape[1] if n is None else int(n)
        if descriptors is None:
            descriptors = torch.zeros((max(n, 1), self.descriptorSize()), dtype=torch.uint8, device=image.device)
        self._check(lib().efx_compute_async(self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0),
                                            keypoints.data_ptr(), keypoints.stride(0) * 4, n, descriptors.data_ptr(),
                                            descriptors.stride(0), _stream_ptr(stream)))
        return descriptors[:n]

    def deviceBytes(self):
        """Device memory the context holds (pyramid, tile headers, corner / survivor arenas, keypoint lists)."""
        return int(lib().efx_device_bytes(self._h))

    def lastCount(self):
        n = C.c_int(0)
        self._check(lib().efx_last_count(self._h, C.byref(n)))
        return n.value

    def overflowEvents(self):
        """Void frames this context has reported (include/efx.h): 0 since round 6 -- a frame of any corner density is complete on
        the first call; kept for callers of the earlier contract."""
        return int(lib().efx_overflow_events(self._h))

    def trackedStreams(self):
        """Diagnostics: streams the context's release waits cover (include/efx.h: efx_tracked_streams)."""
        return int(lib().efx_tracked_streams(self._h))

    def lastLevelStats(self):
        st = (LevelStats * 32)()
        nl = C.c_int(0)
        self._check(lib().efx_last_level_stats(self._h, st, 32, C.byref(nl)))
        return [dict(n_candidates=st[i].n_candidates, n_after_nms=st[i].n_after_nms, n_kept=st[i].n_kept)
                for i in range(nl.value)]

    def profileEnable(self, max_launches, stride=1, groups=0x3f):
        """groups: bit 0 fast, 1 harris, 2 nms, 3 select+emit+angle, 4 describe, 5 pyramid (include/efx.h)."""
        self._check(lib().efx_profile_enable(self._h, int(max_launches)))
        self._check(lib().efx_profile_set_stride(self._h, int(stride)))
        self._check(lib().efx_profile_set_groups(self._h, int(groups)))

    def profileRead(self, capacity=65536):
        """(ms, code) arrays of the recorded launches (codes: include/efx.h, efx_profile_enable); call after synchronising."""
        ms = (C.c_float * capacity)()
        lv = (C.c_int * capacity)()
        n = C.c_int(0)
        self._check(lib().efx_profile_read(self._h, ms, lv, capacity, C.byref(n)))
        return np.array(ms[:n.value], dtype=np.float64), np.array(lv[:n.value], dtype=np.int64)

    def levelGeometry(self, rows, cols, level):
        r, c, s = C.c_int(), C.c_int(), C.c_float()
        self._check(lib().efx_level_geometry(self._h, rows, cols, level, C.byref(r), C.byref(c), C.byref(s)))
        return r.value, c.value, s.value

    def copyLevel(self, level, rows, cols, stream=None):
        import torch
        r, c, _ = self.levelGeometry(rows, cols, level)
        out = torch.zeros((r, c), dtype=torch.uint8, device="cuda")
        self._check(lib().efx_copy_level_async(self._h, level, out.data_ptr(), out.stride(0), _stream_ptr(stream)))
        return out

    # ---- synchronous host API ----
    def detect(self, image, capacity=None):
        img = _host_image(image)
        capacity = self.getMaxFeatures() if capacity is None else int(capacity)
        kps = keypoints_array(max(capacity, 1))
        n = C.c_int(0)
        self._check(lib().efx_detect(self._h, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0],
                                     kps.ctypes.data, capacity, C.byref(n)))
        return kps[:n.value].copy()

    def compute(self, image, keypoints):
        img = _host_image(image)
        kps = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((max(len(kps), 1), self.descriptorSize()), dtype=np.uint8)
        self._check(lib().efx_compute(self._h, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0],
                                      kps.ctypes.data, len(kps), desc.ctypes.data, desc.strides[0]))
        return desc[:len(kps)]

    def detectAndCompute(self, image, capacity=None, useProvidedKeypoints=False, mask=None, keypoints=None):
        """Feature2D::detectAndCompute(image, mask, keypoints, descriptors, useProvidedKeypoints) on host arrays."""
        img = _host_image(image)
        capacity = self.getMaxFeatures() if capacity is None else int(capacity)
        m = None
        if mask is not None:
            m = _host_image(mask)
            if m.shape != img.shape:
                raise EfxError(-1, "mask must have the size of the image")
        if useProvidedKeypoints:
            if keypoints is None:
                raise EfxError(-1, "useProvidedKeypoints needs keypoints")
            kps = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
            n = C.c_int(len(kps))
            desc = np.zeros((max(len(kps), 1), self.descriptorSize()), dtype=np.uint8)
        else:
            kps = keypoints_array(max(capacity, 1))
            n = C.c_int(0)
            desc = np.zeros((max(capacity, 1), self.descriptorSize()), dtype=np.uint8)
        self._check(lib().efx_detect_and_compute_ex(self._h, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0],
                                                    m.ctypes.data if m is not None else None, m.strides[0] if m is not None else 0,
                                                    kps.ctypes.data, desc.ctypes.data, desc.strides[0], capacity, C.byref(n),
                                                    1 if useProvidedKeypoints else 0))
        return kps[:n.value].copy(), desc[:n.value].copy()

    @staticmethod
    def convert(gpu_keypoints, n=None):
        """5 x N matrix (CUDA tensor or ndarray) -> keypoint records (cuda_efficient_features.cpp:323-349)."""
        arr = gpu_keypoints.detach().cpu().numpy() if hasattr(gpu_keypoints, "detach") else np.asarray(gpu_keypoints)
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        n = arr.shape[1] if n is None else int(n)
        out = keypoints_array(max(n, 1))
        rc = lib().efx_convert(arr.ctypes.data, arr.strides[0], n, out.ctypes.data)
        if rc != EFX_OK:
            raise EfxError(rc, "efx_convert")
        return out[:n].copy()


class _Describer:
    SIZE_512_BITS, SIZE_256_BITS = 100, 101

    def __init__(self, kind, scale, nbits):
        self._h = C.c_void_p()
        fn = lib().efx_bad_create if kind == "bad" else lib().efx_hashsift_create
        rc = fn(C.c_float(scale), int(nbits), C.byref(self._h))
        if rc != EFX_OK:
            self._h = C.c_void_p()
            raise EfxError(rc, lib().efx_describer_last_error(None).decode())

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().efx_describer_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc != EFX_OK:
            raise EfxError(rc, lib().efx_describer_last_error(self._h).decode())

    def descriptorSize(self): return lib().efx_describer_descriptor_size(self._h)
    def descriptorType(self): return 0
    def defaultNorm(self): return 6

    def compute(self, image, keypoints):
        """image: host uint8, H x W (8UC1), H x W x 3 (BGR) or H x W x 4 (BGRA) as the CPU describers accept
        (bad.cpp:268-281); keypoints: KEYPOINT_DTYPE records or (n,4) float32 {x,y,size,angle}."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or not (img.ndim == 2 or (img.ndim == 3 and img.shape[2] in (1, 3, 4))):
            raise EfxError(-1, "Image should be 8UC1, 8UC3 or 8UC4")
        channels = 1 if img.ndim == 2 else img.shape[2]
        kps = np.asarray(keypoints)
        if kps.dtype != KEYPOINT_DTYPE:
            k4 = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 4)
            kps = keypoints_array(len(k4))
            kps["x"], kps["y"], kps["size"], kps["angle"] = k4[:, 0], k4[:, 1], k4[:, 2], k4[:, 3]
        kps = np.ascontiguousarray(kps)
        desc = np.zeros((max(len(kps), 1), self.descriptorSize()), dtype=np.uint8)
        self._check(lib().efx_describer_compute_color(self._h, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], channels,
                                                      kps.ctypes.data, len(kps), desc.ctypes.data, desc.strides[0]))
        return desc[:len(kps)]

    def computeAsync(self, image, keypoints, n=None, descriptors=None, max_size=0.0, stream=None):
        """keypoints: 5 x N float32 CUDA tensor (detector layout) or N x 4 float32 CUDA tensor {x,y,size,angle}."""
        import torch
        image = _dev_image(image)
        five = keypoints.dim() == 2 and keypoints.shape[0] == 5 and keypoints.shape[1] != 4
        n = (keypoints.shape[1] if five else keypoints.shape[0]) if n is None else int(n)
        if descriptors is None:
            descriptors = torch.zeros((max(n, 1), self.descriptorSize()), dtype=torch.uint8, device=image.device)
        if five:
            self._check(lib().efx_describer_compute_async(
                self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), keypoints.data_ptr(),
                keypoints.stride(0) * 4, n, descriptors.data_ptr(), descriptors.stride(0), _stream_ptr(stream)))
        else:
            k = keypoints.contiguous()
            self._check(lib().efx_describer_compute_kp4_async(
                self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), k.data_ptr(), n,
                C.c_float(max_size), descriptors.data_ptr(), descriptors.stride(0), _stream_ptr(stream)))
        return descriptors[:n]


class BAD(_Describer):
    """cv::cuda::BAD (cuda_efficient_descriptors.h:66-90)."""
    def __init__(self, scaleFactor, nbits=_Describer.SIZE_256_BITS):
        super().__init__("bad", scaleFactor, nbits)

    @staticmethod
    def create(scaleFactor, nbits=_Describer.SIZE_256_BITS):
        return BAD(scaleFactor, nbits)


class HashSIFT(_Describer):
    """cv::cuda::HashSIFT (cuda_efficient_descriptors.h:101-121)."""
    def __init__(self, croppingScale, nbits=_Describer.SIZE_256_BITS):
        super().__init__("hashsift", croppingScale, nbits)

    @staticmethod
    def create(croppingScale, nbits=_Describer.SIZE_256_BITS):
        return HashSIFT(croppingScale, nbits)

    def debug(self, image, kp4, max_size=0.0, stream=None):
        """Returns (129-vectors n x 129, pre-threshold projections n x nbits) as CUDA tensors."""
        import torch
        image = _dev_image(image)
        k = kp4.contiguous()
        n = k.shape[0]
        nbits = self.descriptorSize() * 8
        resp = torch.zeros((max(n, 1), 129), dtype=torch.float32, device=image.device)
        T = torch.zeros((max(n, 1), nbits), dtype=torch.float32, device=image.device)
        self._check(lib().efx_describer_hashsift_debug_async(
            self._h, image.data_ptr(), image.shape[0], image.shape[1], image.stride(0), k.data_ptr(), n,
            C.c_float(max_size), resp.data_ptr(), T.data_ptr(), _stream_ptr(stream)))
        return resp[:n], T[:n]



def icAngles(image, keypoints, patch_size):
    """ICAngles of samples/hpatches_description.cpp:128-162 on host data: returns a copy of the keypoint records with
    the angle filled (computed on the device)."""
    img = _host_image(image)
    kps = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE).copy()
    rc = lib().efx_ic_angles(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], kps.ctypes.data, len(kps), int(patch_size))
    if rc != EFX_OK:
        raise EfxError(rc, lib().efx_last_error(None).decode())
    return kps


def descriptorsToCsv(descriptors):
    """saveDescriptors of samples/hpatches_description.cpp:76-105: the text of the CSV file."""
    d = np.ascontiguousarray(descriptors, dtype=np.uint8)
    n, nbytes = d.shape
    size = lib().efx_descriptors_to_csv(d.ctypes.data, n, nbytes, d.strides[0], None, 0)
    if size < 0:
        raise EfxError(-1, "bad descriptor matrix")
    buf = C.create_string_buffer(int(size) + 1)
    lib().efx_descriptors_to_csv(d.ctypes.data, n, nbytes, d.strides[0], buf, int(size))
    return buf.raw[:size].decode("ascii")


class Batch:
    """efx_detect_and_compute_batch_async with prepared pointer tables: nframes frames of one size, frame i on
    detectors[i % n] / streams[i % n]; the frames of one detector go through ONE launch of every kernel (round 6; up to 16 per
    launch chain).  The tensors are kept referenced here; run() only crosses the ABI once."""

    def __init__(self, detectors, streams, images, keypoints, descriptors, counts, capacity):
        n, f = len(detectors), len(images)
        self._keep = (detectors, streams, images, keypoints, descriptors, counts)
        P = C.c_void_p
        self._ctx = (P * n)(*[d._h for d in detectors])
        self._st = (P * n)(*[P(s.cuda_stream) for s in streams])
        self._img = (P * f)(*[P(t.data_ptr()) for t in images])
        self._kps = (P * f)(*[P(t.data_ptr()) for t in keypoints])
        self._desc = (P * f)(*[P(t.data_ptr()) for t in descriptors]) if descriptors is not None else None
        self._cnt = (P * f)(*[P(t.data_ptr()) for t in counts])
        im = images[0]
        self._args = (n, f, im.shape[0], im.shape[1], im.stride(0), keypoints[0].stride(0) * 4,
                      descriptors[0].stride(0) if descriptors is not None else 0, int(capacity))
        self._det0 = detectors[0]

    def run(self):
        n, f, rows, cols, pitch, kp, dp, cap = self._args
        rc = lib().efx_detect_and_compute_batch_async(self._ctx, self._st, n, self._img, f, rows, cols, pitch, self._kps, kp,
                                                      self._desc, dp, cap, self._cnt)
        if rc != EFX_OK:
            raise EfxError(rc, "batch: " + lib().efx_last_error(self._det0._h).decode())

    def overflowEvents(self):
        """Void frames over all contexts of the batch: always 0 since round 6 (the scratch arenas hold the reference's own 10 %
        candidate cap; include/efx.h); kept for callers of the earlier contract."""
        return sum(d.overflowEvents() for d in self._keep[0])


def trimMemory():
    """Returns the device blocks cached from destroyed / regrown contexts to the driver; bytes released."""
    return int(lib().efx_trim_memory())


def cachedBytes():
    return int(lib().efx_cached_bytes())


def cvtGray(image, out=None, stream=None):
    """cv::cvtColor(BGR2GRAY / BGRA2GRAY) on the device: H x W x 3|4 uint8 CUDA tensor -> H x W uint8 CUDA tensor."""
    import torch
    if not (isinstance(image, torch.Tensor) and image.is_cuda and image.dtype == torch.uint8 and image.dim() == 3
            and image.shape[2] in (3, 4) and image.stride(2) == 1 and image.stride(1) == image.shape[2]):
        raise EfxError(-1, "Image should be 8UC3 or 8UC4 (H x W x C uint8 CUDA tensor, packed pixels)")
    rows, cols, ch = image.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.uint8, device=image.device)
    rc = lib().efx_cvt_gray_async(image.data_ptr(), rows, cols, image.stride(0), ch, out.data_ptr(), out.stride(0), _stream_ptr(stream))
    if rc != EFX_OK:
        raise EfxError(rc, lib().efx_last_error(None).decode())
    return out


def host_alloc(shape):
    """uint8 numpy array in page-locked host memory (cv::cuda::HostMem); keep the returned array alive while in use."""
    nbytes = int(np.prod(shape))
    p = C.c_void_p()
    rc = lib().efx_host_alloc(nbytes, C.byref(p))
    if rc != EFX_OK:
        raise EfxError(rc, lib().efx_last_error(None).decode())
    buf = (C.c_uint8 * nbytes).from_address(p.value)
    arr = np.frombuffer(buf, dtype=np.uint8).reshape(shape)
    _PINNED[arr.__array_interface__["data"][0]] = p
    return arr


def host_free(arr):
    p = _PINNED.pop(arr.__array_interface__["data"][0], None)
    if p is not None:
        lib().efx_host_free(p)


_PINNED = {}


class Uploader:
    """Double-buffered host -> device input stage (getInputMat upload, cuda_efficient_features.cpp:71-84)."""

    def __init__(self):
        self._h = C.c_void_p()
        rc = lib().efx_uploader_create(C.byref(self._h))
        if rc != EFX_OK:
            self._h = C.c_void_p()
            raise EfxError(rc, lib().efx_uploader_last_error(None).decode())

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().efx_uploader_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def upload(self, image, stream=None):
        """host uint8 H x W [x 3|4] -> (device pointer, pitch, rows, cols) of the gray frame, ordered on `stream`."""
        img = image
        if img.dtype != np.uint8 or not (img.ndim == 2 or (img.ndim == 3 and img.shape[2] in (3, 4))) or img.strides[-1] != 1:
            raise EfxError(-1, "Image should be 8UC1, 8UC3 or 8UC4")
        ch = 1 if img.ndim == 2 else img.shape[2]
        if img.ndim == 3 and img.strides[1] != ch:
            raise EfxError(-1, "pixels must be packed")
        d = C.c_void_p(); pitch = C.c_size_t()
        rc = lib().efx_upload_gray_async(self._h, img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], ch,
                                         C.byref(d), C.byref(pitch), _stream_ptr(stream))
        if rc != EFX_OK:
            raise EfxError(rc, lib().efx_uploader_last_error(self._h).decode())
        return d.value, pitch.value, img.shape[0], img.shape[1]

    def release(self, d_gray, stream=None):
        """Everything enqueued on `stream` so far is the last reader of frame `d_gray`: recycle its slot only behind it."""
        rc = lib().efx_uploader_release(self._h, d_gray, _stream_ptr(stream))
        if rc != EFX_OK:
            raise EfxError(rc, lib().efx_uploader_last_error(self._h).decode())

    def waitUploaded(self, d_gray):
        """Block until frame `d_gray` has left the host buffer it was uploaded from."""
        rc = lib().efx_uploader_wait_uploaded(self._h, d_gray)
        if rc != EFX_OK:
            raise EfxError(rc, lib().efx_uploader_last_error(self._h).decode())


class BFMatcher:
    """cv::BFMatcher(NORM_HAMMING[, crossCheck]) on the device (samples/sample_feature_matching.cpp:99-101,
    samples/sample_image_sequence.cpp:81,114-115).  Descriptors are N x 32 / N x 64 uint8 CUDA tensors."""
    NORM_HAMMING = 6

    def __init__(self, normType=6, crossCheck=False):
        if normType != 6:
            raise EfxError(-2, "only NORM_HAMMING is supported")
        self.crossCheck = bool(crossCheck)
        self._h = C.c_void_p()
        rc = lib().efx_matcher_create(C.byref(self._h))
        if rc != EFX_OK:
            self._h = C.c_void_p()
            raise EfxError(rc, lib().efx_matcher_last_error(None).decode())

    @staticmethod
    def create(normType=6, crossCheck=False):
        return BFMatcher(normType, crossCheck)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().efx_matcher_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc != EFX_OK:
            raise EfxError(rc, lib().efx_matcher_last_error(self._h).decode())

    @staticmethod
    def _desc(t):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.stride(1) == 1
                and t.shape[1] in (32, 64)):
            raise EfxError(-1, "descriptors must be an N x 32 or N x 64 uint8 CUDA tensor")
        return t

    def knnMatch(self, query, train, k=2, stream=None):
        """Returns (idx, dist): nq x 2 int32 CUDA tensors, nearest first (distance = differing bits)."""
        import torch
        if k != 2:
            raise EfxError(-2, "k must be 2")
        q, t = self._desc(query), self._desc(train)
        if q.shape[1] != t.shape[1]:
            raise EfxError(-1, "query and train descriptors differ in size")
        idx = torch.full((max(q.shape[0], 1), 2), -1, dtype=torch.int32, device=q.device)
        dist = torch.full((max(q.shape[0], 1), 2), -1, dtype=torch.int32, device=q.device)
        self._check(lib().efx_match_knn2_async(self._h, q.data_ptr(), q.stride(0), q.shape[0], t.data_ptr(), t.stride(0),
                                               t.shape[0], q.shape[1], idx.data_ptr(), dist.data_ptr(), _stream_ptr(stream)))
        return idx[:q.shape[0]], dist[:q.shape[0]]

    def match(self, query, train, stream=None):
        """crossCheck matcher: (trainIdx per query or -1, distance).  Without crossCheck: the nearest neighbour."""
        import torch
        q, t = self._desc(query), self._desc(train)
        if not self.crossCheck:
            idx, dist = self.knnMatch(q, t, 2, stream)
            return idx[:, 0].contiguous(), dist[:, 0].contiguous()
        m = torch.full((max(q.shape[0], 1),), -1, dtype=torch.int32, device=q.device)
        d = torch.full((max(q.shape[0], 1),), -1, dtype=torch.int32, device=q.device)
        self._check(lib().efx_match_crosscheck_async(self._h, q.data_ptr(), q.stride(0), q.shape[0], t.data_ptr(), t.stride(0),
                                                     t.shape[0], q.shape[1], m.data_ptr(), d.data_ptr(), _stream_ptr(stream)))
        return m[:q.shape[0]], d[:q.shape[0]]


def unpack_keypoints(kps):
    """(5,N) float32 ndarray -> dict of x, y (int16), response, angle, octave (int32), size."""
    kps = np.ascontiguousarray(kps, dtype=np.float32)
    loc = kps[0].view(np.uint32)
    x = (loc & 0xFFFF).astype(np.uint16).view(np.int16)
    y = (loc >> 16).astype(np.uint16).view(np.int16)
    return dict(x=x, y=y, response=kps[1].copy(), angle=kps[2].copy(), octave=kps[3].view(np.int32).copy(),
                size=kps[4].copy())

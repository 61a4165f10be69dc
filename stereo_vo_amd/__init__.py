"""stereo_vo_amd -- MI355X-native stereo visual-odometry hot path behind the API of
rso::CStereoOdometryEstimator::processNewImagePair (libstereo-odometry.h:157).

Importing the package loads nothing native; `stereo_vo_amd.hip.lib()` loads the HIP C-ABI library
(stereo_vo_amd/libsvo_hip.so) and raises if it is missing -- there is no CPU fallback.
"""

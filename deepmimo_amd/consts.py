"""Names shared with the reference at the API boundary (deepmimo/consts.py).

Only the string keys a user of ``Dataset`` / ``ChannelGenParameters`` can observe are kept:
matrix names (consts.py:187-228), channel-parameter keys (:234-254) and the alias table
(:261-322).  Keeping them verbatim is what makes this package a drop-in for the channel path.
"""
import numpy as np

VERSION = "4.0.0a3+mi355x"

FP_TYPE = np.float32            # storage precision of the ray matrices (consts.py:65)
MAX_PATHS = 25                  # consts.py:180

INTERACTION_LOS = 0             # consts.py:173

# fundamental matrices
POWER_PARAM_NAME = "power"
PHASE_PARAM_NAME = "phase"
DELAY_PARAM_NAME = "delay"
AOA_AZ_PARAM_NAME = "aoa_az"
AOA_EL_PARAM_NAME = "aoa_el"
AOD_AZ_PARAM_NAME = "aod_az"
AOD_EL_PARAM_NAME = "aod_el"
RX_POS_PARAM_NAME = "rx_pos"
TX_POS_PARAM_NAME = "tx_pos"
INTERACTIONS_PARAM_NAME = "inter"
INTERACTIONS_POS_PARAM_NAME = "inter_pos"
DOPPLER_VEL_PARAM_NAME = "doppler_vel"      # not in v4; v3 'Doppler_vel' (deepmimo_v3/consts.py:63)
DOPPLER_ACC_PARAM_NAME = "doppler_acc"      # v3 'Doppler_acc' (deepmimo_v3/consts.py:64)

RAY_FIELDS = (POWER_PARAM_NAME, PHASE_PARAM_NAME, DELAY_PARAM_NAME, AOA_AZ_PARAM_NAME, AOA_EL_PARAM_NAME,
              AOD_AZ_PARAM_NAME, AOD_EL_PARAM_NAME, INTERACTIONS_PARAM_NAME)

# computed
CHANNEL_PARAM_NAME = "channel"
CH_PARAMS_PARAM_NAME = "ch_params"
LOS_PARAM_NAME = "los"
NUM_PATHS_PARAM_NAME = "num_paths"
PWR_LINEAR_PARAM_NAME = "power_linear"
PATHLOSS_PARAM_NAME = "pathloss"
DIST_PARAM_NAME = "distance"
N_UE_PARAM_NAME = "n_ue"
INTER_INT_PARAM_NAME = "inter_int"
NUM_INTERACTIONS_PARAM_NAME = "num_interactions"
INTER_STR_PARAM_NAME = "inter_str"

AOA_AZ_ROT_PARAM_NAME = "_aoa_az_rot"
AOA_EL_ROT_PARAM_NAME = "_aoa_el_rot"
AOD_AZ_ROT_PARAM_NAME = "_aod_az_rot"
AOD_EL_ROT_PARAM_NAME = "_aod_el_rot"
AOD_EL_FOV_PARAM_NAME = "_aod_el_rot_fov"
AOD_AZ_FOV_PARAM_NAME = "_aod_az_rot_fov"
AOA_EL_FOV_PARAM_NAME = "_aoa_el_rot_fov"
AOA_AZ_FOV_PARAM_NAME = "_aoa_az_rot_fov"
FOV_MASK_PARAM_NAME = "_fov_mask"
PWR_LINEAR_ANT_GAIN_PARAM_NAME = "_power_linear_ant_gain"

RT_PARAMS_PARAM_NAME = "rt_params"
RT_PARAM_FREQUENCY = "frequency"
SCENE_PARAM_NAME = "scene"
MATERIALS_PARAM_NAME = "materials"
LOAD_PARAMS_PARAM_NAME = "load_params"
TXRX_PARAM_NAME = "txrx_sets"

# channel generation parameters
PARAMSET_POLAR_EN = "enable_dual_polar"
PARAMSET_DOPPLER_EN = "enable_doppler"
PARAMSET_FD_CH = "freq_domain"
PARAMSET_NUM_PATHS = "num_paths"
PARAMSET_OFDM = "ofdm"
PARAMSET_OFDM_SC_NUM = "subcarriers"
PARAMSET_OFDM_SC_SAMP = "selected_subcarriers"
PARAMSET_OFDM_BANDWIDTH = "bandwidth"
PARAMSET_OFDM_LPF = "rx_filter"
PARAMSET_ANT_BS = "bs_antenna"
PARAMSET_ANT_UE = "ue_antenna"
PARAMSET_ANT_SHAPE = "shape"
PARAMSET_ANT_SPACING = "spacing"
PARAMSET_ANT_ROTATION = "rotation"
PARAMSET_ANT_RAD_PAT = "radiation_pattern"
PARAMSET_ANT_RAD_PAT_VALS = ["isotropic", "halfwave-dipole"]

DATASET_ALIASES = {
    "los_status": LOS_PARAM_NAME,
    "ch": CHANNEL_PARAM_NAME, "chs": CHANNEL_PARAM_NAME, "channels": CHANNEL_PARAM_NAME,
    "channel_params": CH_PARAMS_PARAM_NAME,
    "pwr": POWER_PARAM_NAME, "powers": POWER_PARAM_NAME,
    "lin_pwr": PWR_LINEAR_PARAM_NAME, "linear_power": PWR_LINEAR_PARAM_NAME, "pwr_lin": PWR_LINEAR_PARAM_NAME,
    "pwr_ant_gain": PWR_LINEAR_ANT_GAIN_PARAM_NAME,
    "ue_pos": RX_POS_PARAM_NAME, "rx_loc": RX_POS_PARAM_NAME, "rx_position": RX_POS_PARAM_NAME,
    "rx_locations": RX_POS_PARAM_NAME,
    "bs_pos": TX_POS_PARAM_NAME, "tx_loc": TX_POS_PARAM_NAME, "tx_position": TX_POS_PARAM_NAME,
    "tx_locations": TX_POS_PARAM_NAME,
    "pl": PATHLOSS_PARAM_NAME, "path_loss": PATHLOSS_PARAM_NAME,
    "dist": DIST_PARAM_NAME, "dists": DIST_PARAM_NAME,
    "aoa_phi": AOA_AZ_PARAM_NAME, "aoa_theta": AOA_EL_PARAM_NAME,
    "aod_phi": AOD_AZ_PARAM_NAME, "aod_theta": AOD_EL_PARAM_NAME,
    "n_paths": NUM_PATHS_PARAM_NAME,
    "toa": DELAY_PARAM_NAME, "time_of_arrival": DELAY_PARAM_NAME,
    "bounce_type": INTERACTIONS_PARAM_NAME, "interactions": INTERACTIONS_PARAM_NAME,
    "bounce_pos": INTERACTIONS_POS_PARAM_NAME, "interaction_positions": INTERACTIONS_POS_PARAM_NAME,
    "interaction_locations": INTERACTIONS_POS_PARAM_NAME,
    "tx_rx": TXRX_PARAM_NAME,
}

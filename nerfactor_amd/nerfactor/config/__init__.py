"""Default configurations, key-compatible with the reference's .ini files
(nerfactor/config/*.ini: one [DEFAULT] section, values are strings read with
config.get*/getint/getfloat/getboolean at call time).  Generated programmatically; use
`make_config(name, **overrides)` or `write_ini(name, path)` to materialise a file for
`--config`."""
import configparser

_COMMON_TRAIN = dict(
    no_batch='True', cache='True', loss='l2', lr_decay_steps='500_000', lr_decay_rate='0.1',
    clipnorm='-1', clipvalue='-1', vis_train_batches='4', keep_recent_epochs='-1',
    overwrite='False', xname='lr{lr}', imh='512', near='2', far='6', ndc='False', white_bg='True')

_SURFACE_MLP = dict(
    mlp_chunk='65536', mlp_width='128', mlp_depth='4', mlp_skip_at='2', pos_enc='True',
    n_freqs_xyz='10', n_freqs_ldir='4', n_freqs_vdir='4', n_rays_per_step='1024', light_h='16',
    xyz_jitter_std='0.01', smooth_use_l1='True')

CONFIGS = {
    'nerf': dict(
        _COMMON_TRAIN, dataset='nerf', model='nerf', lr='1e-4', epochs='2_000',
        ckpt_period='100', vali_period='100', vali_batches='8', use_views='True', pos_enc='True',
        n_freqs_xyz='10', n_freqs_view='4', n_rays_per_step='1024', n_samples_coarse='64',
        n_samples_fine='128', lin_in_disp='False', perturb='True', noise_std='0.',
        accu_chunk='65536', mlp_chunk='65536', mlp_width='256', enc_depth='8', enc_skip_at='4',
        enc_width='256', act='relu'),
    'shape': dict(
        _COMMON_TRAIN, **_SURFACE_MLP, dataset='nerf_shape', model='shape', lr='1e-2',
        epochs='200', ckpt_period='100', vali_period='100', vali_batches='4',
        normal_loss_weight='1', lvis_loss_weight='1'),
    'nerfactor': dict(
        _COMMON_TRAIN, **_SURFACE_MLP, dataset='nerf_shape', model='nerfactor', lr='5e-3',
        epochs='100', ckpt_period='10', vali_period='10', vali_batches='4',
        use_nerf_alpha='False', shape_mode='finetune', nerf_shape_respect='0.1',
        normal_loss_weight='0.1', lvis_loss_weight='0.1', normal_smooth_weight='0.05',
        lvis_smooth_weight='0.05', albedo_slope='0.77', albedo_bias='0.03', pred_brdf='True',
        default_z='0.1', albedo_smooth_weight='0.05', brdf_smooth_weight='0.01',
        learned_brdf_scale='1', light_init_max='1', light_tv_weight='5e-6',
        light_achro_weight='0', linear2srgb='True'),
    'brdf': dict(
        no_batch='True', cache='True', loss='l2', lr='1e-2', lr_decay_steps='500_000',
        lr_decay_rate='0.1', clipnorm='-1', clipvalue='-1', dataset='brdf_merl', model='brdf',
        epochs='50_000', ckpt_period='1_000', vali_period='1_000', vali_batches='4',
        vis_train_batches='4', keep_recent_epochs='-1', overwrite='False', xname='lr{lr}',
        shuffle_buffer_size='65536', pos_enc='True', n_freqs='2', n_rays_per_step='1024',
        z_dim='3', z_gauss_mean='0.', z_gauss_std='0.01', normalize_z='False',
        loss_transform='log', mlp_chunk='65536', mlp_width='128', mlp_depth='4', mlp_skip_at='2'),
}
CONFIGS['nerfactor_microfacet'] = dict(
    {k: v for k, v in CONFIGS['nerfactor'].items()
     if k not in ('default_z', 'learned_brdf_scale')},
    model='nerfactor_microfacet', rough_min='0.1', default_rough='0.3', fresnel_f0='0.04',
    brdf_smooth_weight='0')
# the reference's remaining shipped .ini files (config/*.ini) differ from the ones above in a few keys each
CONFIGS['shape_mvs'] = dict(CONFIGS['shape'], dataset='mvs_shape', xyz_scale='1e-3')              # DTU scenes have huge XYZs
CONFIGS['nerfactor_mvs'] = dict(CONFIGS['nerfactor'], dataset='mvs_shape', xyz_scale='1e-3')
CONFIGS['nerfactor_no_geom_opt'] = dict(CONFIGS['nerfactor'], shape_mode='nerf')                 # (the file's name; its outroot says "no_geom_refine")
CONFIGS['nerfactor_no_geom_pretrain'] = dict(CONFIGS['nerfactor'], shape_mode='scratch')
CONFIGS['nerfactor_no_smooth'] = dict(CONFIGS['nerfactor'], normal_smooth_weight='0', lvis_smooth_weight='0',
                                      albedo_smooth_weight='0', brdf_smooth_weight='0')


def make_config(name, **overrides):
    cfg = configparser.ConfigParser()
    cfg['DEFAULT'] = dict(CONFIGS[name])
    for k, v in overrides.items():
        cfg['DEFAULT'][k] = str(v)
    return cfg


def write_ini(name, path, **overrides):
    with open(path, 'w') as h:
        make_config(name, **overrides).write(h)

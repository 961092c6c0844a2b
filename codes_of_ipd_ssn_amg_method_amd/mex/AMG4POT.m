function [zeta,itamg,resamg,info] = AMG4POT(prob_data,amg_options,str)
% Drop-in shim (Class2/AMG4POT.m:1): str = 'amg' (Hybrid_AMG) or 'twogrid' (Hybrid_twogrid).
if strcmp(str,'amg')
    [zeta,itamg,resamg,info] = ipd_mex('AMG4POT', prob_data, amg_options);
else
    [zeta,itamg,resamg,info] = ipd_mex('AMG4POT_twogrid', prob_data, amg_options);
end
end

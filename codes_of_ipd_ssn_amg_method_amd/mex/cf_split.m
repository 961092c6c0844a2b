function [indC,indF] = cf_split(S)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[indC,indF] = ipd_mex('cf_split', S);
end

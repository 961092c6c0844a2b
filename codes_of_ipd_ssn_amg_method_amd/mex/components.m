function [blocks,sizes,p,r] = components(A)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[blocks,sizes,p,r] = ipd_mex('components', A);
end

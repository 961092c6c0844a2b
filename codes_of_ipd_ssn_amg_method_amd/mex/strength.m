function S = strength(varargin)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[S] = ipd_mex('strength', varargin{:});
end
